from .block import MI355XHunyuanDualBlock, MI355XHunyuanSingleBlock  # noqa: F401
from .model import HunyuanVideoTransformerConfig, MI355XHunyuanVideoTransformer3DModel, rotary_tables  # noqa: F401
from .specification import MI355XHunyuanVideoModelSpecification, MI355XHunyuanVideoSpecOps  # noqa: F401
from .trainer import MI355XHunyuanVideoSFTStep  # noqa: F401
