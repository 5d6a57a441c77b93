from .block import MI355XHunyuanDualBlock, MI355XHunyuanSingleBlock  # noqa: F401
