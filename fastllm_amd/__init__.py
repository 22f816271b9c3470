"""fastllm_amd -- MI355X (gfx950) forward-pass backend for FastLLM behind a C ABI.

The product is fastllm_amd/lib/libfastllm_mi355x.so (sources: fastllm_amd/csrc, ABI:
include/fastllm_mi355x.h).  This Python package is the thin ctypes harness the tests and
bench.py drive it with; it contains no arithmetic and no CPU fallback.
"""
from .binding import (Batch, Cache, FastLLMError, Model, abi_version, comm_unique_id, device_count, lib, op_attention, op_linear, op_sample,  # noqa: F401
                      tp_slice, tune, reload_env, library_loaded)
from .configs import MODEL_CONFIGS  # noqa: F401
