from .transformer import LTXTransformerConfig, MI355XLTXVideoTransformer3DModel  # noqa: F401
from .specification import MI355XLTXVideoModelSpecification  # noqa: F401
