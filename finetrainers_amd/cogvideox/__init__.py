from .specification import CogVideoXDDIMTables, MI355XCogVideoXSpecOps  # noqa: F401
