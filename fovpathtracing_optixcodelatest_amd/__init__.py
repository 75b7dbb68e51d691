"""MI355X-native foveated path-tracing launch (drop-in for the reference's SampleRenderer::render path)."""
from . import abi, scenes  # noqa: F401
