from .unified_encoder import (EntitySpatialCrossEncoder, UnifiedSpatialCrossEncoderV1,  # noqa: F401
                              UnifiedSpatialCrossEncoderV2)
