"""sceneverse_amd -- MI355X-native GPS hot path (PointNet++ set abstraction + spatial/language
transformer) behind the reference's registry API.  See DESIGN.md / INTEGRATION.md."""

__version__ = "0.1.0"
