"""cpp-paddle-ocr_amd — MI355X-native OCR hot path (det -> cls -> rec) behind a C-ABI.

The directory name mirrors the reference repository and is not an importable identifier;
load it with `importlib` (see `__graft_entry__.load_package()`), which registers it as
`cpp_paddle_ocr_amd`.  Python here is test/bench plumbing only: the product is
`lib/libocr_hip.so` (include/ocr_hip.h) and the C++ host layer under `host/`.
"""
from .binding import *  # noqa: F401,F403
