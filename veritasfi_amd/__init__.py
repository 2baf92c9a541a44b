"""veritasfi_amd -- MI355X-native (gfx950) embedding / retrieval / re-rank hot path for VeritasFi.

Host code is Python; all compute is hand-written HIP behind a C ABI (``include/veritasfi_hip.h``,
``libveritasfi_hip.so``), bound with ctypes.  There is no CPU fallback: importing this package is
cheap and GPU-free, but every operation raises if the HIP library is not built or no GPU is present.

Drop-in names (same signatures as the reference):

* ``FaissRetriever``                     -- ``src/utils/faissRetriever.py``
* ``select_top_chunks(_batch)``, ``get_embeddings``, ``last_token_pool``
                                         -- ``experiments/retriever/step3_mul.py`` / ``continuous_retrieval.py``
* ``compute_similarity_mtx``, ``fuse_and_rank``, ``time_scores``
                                         -- ``src/utils/ensembleRetriever.py:265-281``, ``src/utils/vllmManager.py:443-457``
* ``HipEmbeddings`` / ``HipReranker``   -- ``HuggingFaceEmbeddings`` (``src/utils/ragManager.py:50``) /
                                            ``reranker.compute_score`` (``src/utils/vllmManager.py:451``)
* ``EnsembleRetriever``                  -- ``src/utils/ensembleRetriever.py:19-232`` (candidate gathering around the search)
* ``ShardedRetriever``                   -- row-sharded multi-GPU search
* ``ShardedScorer``                      -- data-parallel re-rank / embed: a replica per rank + one all-gather of the scores
* ``HuggingFaceEmbeddings(model_name=...)`` / ``FlagLLMReranker(name, ...)`` / ``from_config(cfg)``
                                         -- the reference's two constructor calls (``ragManager.py:50``, ``vllmChatService.py:90``) and
                                            its YAML keys (``config/example.yaml``), returning the HIP-backed objects (pretrained.py)
* ``configure()``                        -- optional, explicit: more HIP hardware queues for processes that hold several handles (call before the
                                            first GPU call; importing the package sets nothing)
* ``set_profiler``                       -- routes the reference's stage names ("retrieve", "retrieve_faiss", "retrieve_faiss_ts",
                                            "rerank"; ``src/utils/profiler.py``) out of the drop-in classes (off by default)
"""
import os as _os


def configure(hw_queues: int = 8, warn: bool = True) -> bool:
    """Process-wide runtime settings this package's streams profit from.  EXPLICIT: importing the package changes nothing in the host
    application's environment (until round 6 it set GPU_MAX_HW_QUEUES at import -- a side effect of a drop-in library, and silently
    without effect when the host had touched the GPU first).

    HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  An index keeps two batches in flight on four
    streams of its own, every model handle has one, the caller brings more: past four, streams that are meant to overlap share a queue
    (round 5: a 1M-row search loop 13 % slower in a process that had opened a second index; profiles/r05_hw_queues.log).  The variable is
    read ONCE, when the HIP runtime initialises: call this before anything touches the GPU (before the first torch.cuda call / the
    first index or model of this package).  An explicit setting of the variable by the host wins.  Returns True when the setting will
    take effect; warns (and returns False) when the runtime is already up."""
    import warnings
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return True
    up = False
    try:
        from . import _ffi
        up = _ffi.loaded()
    except Exception:  # noqa: BLE001
        up = False
    try:
        import sys as _sys
        torch = _sys.modules.get("torch")
        up = up or bool(torch is not None and torch.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        pass
    if up:
        if warn:
            warnings.warn("veritasfi_amd.configure(): the HIP runtime is already initialised in this process; GPU_MAX_HW_QUEUES "
                          f"={hw_queues} would not be read any more -- call configure() before the first GPU call", RuntimeWarning, stacklevel=2)
        return False
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    return True


from .index import (DenseIndex, cosine_matrix, cosine_scores, fuse_rank, merge_topk_device,  # noqa: F401
                    merge_topk_packed_device, packed_part_bytes, packed_result_buffer)
from .faiss_retriever import FaissRetriever  # noqa: F401
from .retrieval import get_embeddings, last_token_pool, select_top_chunks, select_top_chunks_batch  # noqa: F401
from .similarity import compute_similarity, compute_similarity_mtx, fuse_and_rank, time_scores  # noqa: F401
from .sharded import ShardedRetriever, ShardedScorer, shard_bounds  # noqa: F401
from .encoder import (HipDecoder, HipDecoderEmbeddings, HipDecoderModel, HipEmbeddings, HipEncoder, HipLLMReranker, HipModel,  # noqa: F401
                      HipReranker, build_llm_reranker_inputs, pack_hf_decoder_weights, pack_hf_weights)
from .rank import rank_chunk  # noqa: F401
from .vision import (HipClipTextEmbeddings, HipClipTextEncoder, HipImageEmbeddings, HipVisionEncoder, pack_hf_clip_text,  # noqa: F401
                     pack_hf_clip_vision)
from .ensemble import EnsembleRetriever  # noqa: F401
from .stages import StageTimer, get_profiler, set_profiler  # noqa: F401
from .pretrained import (FlagLLMReranker, FlagReranker, HuggingFaceEmbeddings, ReplicaSet, from_config, load_embeddings,  # noqa: F401
                         load_reranker, read_sentence_transformers_layout)

__version__ = "0.1.0"
