from .specification import CogVideoXDDIMTables, MI355XCogVideoXSpecOps  # noqa: F401
from .block import MI355XCogVideoXBlock  # noqa: F401
