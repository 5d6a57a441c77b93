from .specification import CogVideoXDDIMTables, MI355XCogVideoXModelSpecification, MI355XCogVideoXSpecOps  # noqa: F401
from .block import MI355XCogVideoXBlock  # noqa: F401
from .model import CogVideoXTransformerConfig, MI355XCogVideoXTransformer3DModel  # noqa: F401
from .trainer import MI355XCogVideoXSFTStep  # noqa: F401
