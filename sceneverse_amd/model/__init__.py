from .build import MODEL_REGISTRY, BaseModel, build_model  # noqa: F401
from .objcls import ObjCls  # noqa: F401
from .openvocab import OpenVocab, OpenVocabPerScene  # noqa: F401
