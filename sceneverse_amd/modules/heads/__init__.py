from .grounding_head import GroundHead, GroundHeadV1  # noqa: F401
from .pretrain_head import OVPretrainHead, PretrainHeadV1  # noqa: F401
from .qa_head import QAHeadV1  # noqa: F401
