from .transformer import LTXTransformerConfig, MI355XLTXVideoTransformer3DModel  # noqa: F401
from .narrow import MI355XNarrowLTXVideoTransformer3DModel, NarrowLayout, build_ltx_transformer  # noqa: F401
from .specification import MI355XLTXVideoModelSpecification  # noqa: F401
