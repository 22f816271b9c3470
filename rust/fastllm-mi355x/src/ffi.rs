//! Raw FFI of `include/fastllm_mi355x.h` (ABI version 2).  One `#[repr(C)]` struct per C struct, one declaration per
//! entry point, same order as the header.  Every entry point returns an `fl_status` (`0` = OK) unless noted; none of
//! them unwinds (the library is an exception barrier), so calling them from Rust is sound.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const FL_ABI_VERSION: c_int = 2;
pub const FL_UNIQUE_ID_BYTES: usize = 128;
pub const FL_IPC_HANDLE_BYTES: usize = 64;

// fl_status
pub const FL_OK: c_int = 0;
pub const FL_ERR_BAD_CONFIG: c_int = -1;
pub const FL_ERR_MISSING_TENSOR: c_int = -2;
pub const FL_ERR_SHAPE_MISMATCH: c_int = -3;
pub const FL_ERR_OOM: c_int = -4;
pub const FL_ERR_HIP: c_int = -5;
pub const FL_ERR_RCCL: c_int = -6;
pub const FL_ERR_SEQ_OVERFLOW: c_int = -7;
pub const FL_ERR_BAD_ARGUMENT: c_int = -8;
pub const FL_ERR_NO_DEVICE: c_int = -9;
pub const FL_ERR_UNSUPPORTED: c_int = -10;

// fl_family (ModelArchitecture::get_family: llama.rs:153, mistral.rs:240, qwen.rs:174)
pub const FL_FAMILY_LLAMA: i32 = 0;
pub const FL_FAMILY_MISTRAL: i32 = 1;
pub const FL_FAMILY_QWEN2: i32 = 2;

// fl_dtype (candle_core::DType subset, dtype_utils.rs:10-23)
pub const FL_DTYPE_F32: i32 = 0;
pub const FL_DTYPE_BF16: i32 = 1;
pub const FL_DTYPE_F16: i32 = 2;

// fl_tp_mode
pub const FL_TP_NONE: i32 = 0;
pub const FL_TP_SINGLE_PROCESS: i32 = 1;
pub const FL_TP_MULTI_PROCESS: i32 = 2;
pub const FL_TP_EMULATED: i32 = 3;

/// `fl_config`: the fields of the reference's ConfigFile / BaseModelConfig (config.rs:6-18).  0 in an optional field
/// = absent from config.json = the reference's default.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct fl_config {
    pub family: i32,
    pub qkv_bias: i32,
    pub hidden_size: i64,
    pub intermediate_size: i64,
    pub vocab_size: i64,
    pub num_hidden_layers: i64,
    pub num_attention_heads: i64,
    pub num_key_value_heads: i64,
    pub max_position_embeddings: i64,
    pub sliding_window: i64,
    pub rms_norm_eps: f64,
    pub rope_theta: f64,
}

/// `fl_tensor`: one entry of initialize_model's `HashMap<String, Tensor>`; borrowed for the call only.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct fl_tensor {
    pub name: *const c_char,
    pub dtype: i32,
    pub ndim: i32,
    pub shape: [i64; 4],
    pub data: *const c_void,
    pub device: i32,
    pub _pad: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct fl_parallel {
    pub mode: i32,
    pub tp_size: i32,
    pub tp_rank: i32,
    pub n_device_ids: i32,
    pub device_ids: *const i32,
    pub unique_id: *const c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct fl_model_info {
    pub cfg: fl_config,
    pub head_dim: i64,
    pub compute_dtype: i32,
    pub tp_size: i32,
    pub weight_bytes_per_token: i64,
    pub kv_bytes_per_position: i64,
    pub hbm_bytes_allocated: i64,
    pub small_collectives: i32,
    pub fused_all_reduce: i32,
    pub rccl_ranks: i32,
    pub _reserved: i32,
}

/// `fl_sampling`: LogitsProcessor::new(seed, Some(temperature), None) (mod.rs:373-374).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct fl_sampling {
    pub temperature: f64,
    pub seed: u64,
    pub draws_done: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct fl_kernel_stat {
    pub name: [c_char; 48],
    pub launches: i64,
    pub total_ms: f64,
    pub bytes: f64,
    pub flops: f64,
}

/// Opaque handles.
#[repr(C)]
pub struct fl_model {
    _private: [u8; 0],
}
#[repr(C)]
pub struct fl_cache {
    _private: [u8; 0],
}
#[repr(C)]
pub struct fl_batch {
    _private: [u8; 0],
}

extern "C" {
    pub fn fl_abi_version() -> c_int;
    /// Thread-local message of the last failure on this thread; never null.
    pub fn fl_last_error() -> *const c_char;
    pub fn fl_device_count(count: *mut c_int) -> c_int;
    pub fn fl_comm_unique_id(out: *mut c_void) -> c_int;

    pub fn fl_model_create(
        cfg: *const fl_config,
        tensors: *const fl_tensor,
        n_tensors: usize,
        compute_dtype: i32,
        par: *const fl_parallel,
        out: *mut *mut fl_model,
    ) -> c_int;
    pub fn fl_comm_ipc_export(m: *mut fl_model, handle_out: *mut c_void) -> c_int;
    pub fn fl_comm_ipc_connect(m: *mut fl_model, handles: *const c_void) -> c_int;
    pub fn fl_model_retain(m: *mut fl_model);
    pub fn fl_model_release(m: *mut fl_model);
    pub fn fl_model_get_info(m: *const fl_model, out: *mut fl_model_info) -> c_int;

    pub fn fl_cache_create(m: *mut fl_model, max_seq: usize, out: *mut *mut fl_cache) -> c_int;
    pub fn fl_cache_reset(c: *mut fl_cache);
    pub fn fl_cache_len(c: *const fl_cache) -> usize;
    pub fn fl_cache_capacity(c: *const fl_cache) -> usize;
    pub fn fl_cache_destroy(c: *mut fl_cache);

    pub fn fl_forward(m: *mut fl_model, c: *mut fl_cache, ids: *const u32, t: usize, pos: usize, logits_out: *mut f32) -> c_int;
    pub fn fl_forward_argmax(m: *mut fl_model, c: *mut fl_cache, ids: *const u32, t: usize, pos: usize, token_out: *mut u32) -> c_int;
    pub fn fl_decode_greedy(
        m: *mut fl_model,
        c: *mut fl_cache,
        first_token: u32,
        pos: usize,
        n_steps: usize,
        eos: i64,
        tokens_out: *mut u32,
        n_out: *mut usize,
    ) -> c_int;
    pub fn fl_forward_sample(
        m: *mut fl_model,
        c: *mut fl_cache,
        ids: *const u32,
        t: usize,
        pos: usize,
        sampling: *const fl_sampling,
        token_out: *mut u32,
    ) -> c_int;
    pub fn fl_decode_sample(
        m: *mut fl_model,
        c: *mut fl_cache,
        first_token: u32,
        pos: usize,
        n_steps: usize,
        eos: i64,
        sampling: *const fl_sampling,
        tokens_out: *mut u32,
        n_out: *mut usize,
    ) -> c_int;

    pub fn fl_batch_create(m: *mut fl_model, caches: *const *mut fl_cache, n: usize, out: *mut *mut fl_batch) -> c_int;
    pub fn fl_batch_destroy(b: *mut fl_batch);
    pub fn fl_batch_replace(b: *mut fl_batch, slot: usize, cache: *mut fl_cache) -> c_int;
    pub fn fl_batch_forward(b: *mut fl_batch, tokens: *const u32, pos: *const usize, logits_out: *mut f32, argmax_out: *mut u32) -> c_int;
    pub fn fl_batch_decode(
        b: *mut fl_batch,
        first_tokens: *const u32,
        pos: *const usize,
        n_steps: usize,
        eos: i64,
        sampling: *const fl_sampling,
        tokens_out: *mut u32,
        n_out: *mut usize,
    ) -> c_int;
    pub fn fl_batch_decode_each(
        b: *mut fl_batch,
        first_tokens: *const u32,
        pos: *const usize,
        n_steps: usize,
        eos: *const i64,
        sampling: *const fl_sampling,
        tokens_out: *mut u32,
        n_out: *mut usize,
    ) -> c_int;

    pub fn fl_synchronize(m: *mut fl_model) -> c_int;
    pub fn fl_tp_slice(cfg: *const fl_config, tensor_name: *const c_char, tp_rank: i32, tp_size: i32, out: *mut i64) -> c_int;

    pub fn fl_profile_begin(m: *mut fl_model) -> c_int;
    pub fn fl_profile_end(m: *mut fl_model, stats: *mut fl_kernel_stat, cap: usize, n_stats: *mut usize) -> c_int;
    pub fn fl_comm_probe(m: *mut fl_model, form: i32, n: i64, iters: i32, us_per_call: *mut f64) -> c_int;
    pub fn fl_comm_selftest(m: *mut fl_model, n: i64, ok: *mut i32) -> c_int;
    pub fn fl_tune(key: *const c_char, value: c_int) -> c_int;
    pub fn fl_op_linear(
        x: *const c_void,
        w: *const c_void,
        bias: *const f32,
        t: i64,
        n: i64,
        k: i64,
        dtype: i32,
        epilogue: i32,
        y: *mut f32,
        iters: i32,
        ms_out: *mut f64,
    ) -> c_int;
    pub fn fl_op_sample(logits: *const f32, v: i64, sampling: *const fl_sampling, n_draws: i64, tokens_out: *mut u32) -> c_int;
    pub fn fl_op_attention(
        q: *const c_void,
        k: *const c_void,
        v: *const c_void,
        t: i64,
        s_past: i64,
        h: i64,
        hkv: i64,
        d: i64,
        window: i64,
        kernel: i32,
        nsplit: i32,
        out: *mut f32,
    ) -> c_int;
}
