from .block import MI355XHunyuanSingleBlock  # noqa: F401
