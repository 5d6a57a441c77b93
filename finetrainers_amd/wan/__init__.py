from .block import MI355XWanBlock, WanBlockLayout  # noqa: F401
from .fsdp import ParameterSharder  # noqa: F401
from .model import MI355XWanTransformer3DModel, WanTransformerConfig, rotary_tables  # noqa: F401
from .specification import MI355XWanModelSpecification, MI355XWanSpecOps  # noqa: F401
from .trainer import MI355XWanFullFinetuneStep  # noqa: F401
