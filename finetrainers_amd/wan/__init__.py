from .block import MI355XWanBlock, WanBlockLayout  # noqa: F401
