"""kjarni_amd -- MI355X-native encoder-inference core behind the Kjarni C ABI.

Python here is a thin ctypes mirror of the reference's Python binding
(crates/kjarni-ffi/bindings/python/kjarni); all compute is in
kjarni_amd/lib/libkjarni_ffi.so (hand-written gfx950 HIP kernels + C++ host).
"""
from ._ffi import KjarniDevice, KjarniError, KjarniException, lib  # noqa: F401
from .encoder import (COSINE_SEGMENT, COSINE_VECTOR_STORE, MASK_AUTO, MASK_NEG_1E9, MASK_NEG_INF,  # noqa: F401
                      POOL_CLS, POOL_LAST_TOKEN, POOL_MAX, POOL_MEAN, HipEncoder, HipEncoderGroup, cosine_search,
                      device_count)

from .classifier import Classifier  # noqa: F401,E402
from .embedder import Embedder  # noqa: F401,E402
from .reranker import Reranker, RerankResult  # noqa: F401,E402
from .searcher import Searcher, search_keywords  # noqa: F401,E402
from .indexer import CancelToken, Indexer, index_delete, index_info  # noqa: F401,E402
from .transcriber import HipWhisper, Transcriber  # noqa: F401,E402
from .decoder import HipDecoder  # noqa: F401,E402
from .chat import BpeTokenizer, Chat, ChatConversation, GenerationConfig  # noqa: F401,E402
from .tokenizer import Tokenizer  # noqa: F401,E402

__version__ = "0.1.0"
