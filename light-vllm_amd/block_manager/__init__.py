from .interfaces import AllocStatus, BlockSpaceManager  # noqa: F401
