"""CPU oracle for the VeritasFi retrieval hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  Nothing under ``veritasfi_amd/`` imports it, and the product path
raises when its HIP library is missing rather than falling back to anything here.

Two layers:

* ``oracle.ref_numpy`` -- a NumPy restatement of what the reference literally runs
  (``sklearn.cosine_similarity`` + ``np.argsort``: ``experiments/retriever/step3_mul.py:233-289``,
  ``continuous_retrieval.py:154-167``).  It is the bridge to the real reference import: the
  golden vectors in ``tests/golden/`` were produced by importing those two reference modules
  (``tools/gen_golden.py``), and this restatement is checked against them.
* ``oracle.canonical`` -- ctypes binding of ``libvf_oracle.so`` (``vf_oracle.c``), the same
  algorithm with a FIXED fp32 summation order ("canonical score") so that a GPU
  implementation can be compared bit-for-bit: ids, rank order and score bits.
"""
from . import ref_numpy  # noqa: F401
from . import canonical  # noqa: F401
