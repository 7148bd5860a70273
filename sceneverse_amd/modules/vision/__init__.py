from .pcd_openvocab_encoder import PointOpenVocabEncoder  # noqa: F401
from .obj_cls_encoder import ObjClsEncoder  # noqa: F401
