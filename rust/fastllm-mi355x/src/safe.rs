//! Safe layer over [`crate::ffi`]: RAII handles and `Result`s.  No candle types here -- tensors come in as
//! [`TensorView`]s (name, dtype, shape, bytes) and logits go out as `Vec<f32>`; the trait glue that converts candle's
//! `Tensor` to and from these lives in the patched reference tree (`src/models/mi355x.rs`).
use std::ffi::{CStr, CString};
use std::fmt;
use std::os::raw::c_void;
use std::ptr;

use crate::ffi;

/// A failed library call: the status code and `fl_last_error()` of the calling thread.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "fastllm_mi355x error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}
pub type Result<T> = std::result::Result<T, Error>;

fn check(rc: i32) -> Result<()> {
    if rc == ffi::FL_OK {
        return Ok(());
    }
    // SAFETY: fl_last_error never returns null and the buffer is thread-local to this thread.
    let message = unsafe { CStr::from_ptr(ffi::fl_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum Family {
    Llama = ffi::FL_FAMILY_LLAMA,
    Mistral = ffi::FL_FAMILY_MISTRAL,
    Qwen2 = ffi::FL_FAMILY_QWEN2,
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum DType {
    F32 = ffi::FL_DTYPE_F32,
    BF16 = ffi::FL_DTYPE_BF16,
    F16 = ffi::FL_DTYPE_F16,
}

/// The fields of the reference's `BaseModelConfig` (src/models/config.rs:6-18); `None` = absent from config.json, the
/// library applies the reference's default (llama.rs:39-47, mistral.rs:97-139, qwen.rs:45-49).
#[derive(Clone, Debug)]
pub struct Config {
    pub family: Family,
    pub hidden_size: usize,
    pub intermediate_size: usize,
    pub vocab_size: usize,
    pub num_hidden_layers: usize,
    pub num_attention_heads: usize,
    pub num_key_value_heads: Option<usize>,
    pub rms_norm_eps: f64,
    pub rope_theta: Option<f64>,
    pub max_position_embeddings: Option<usize>,
    pub sliding_window: Option<usize>,
}

impl Config {
    pub fn to_ffi(&self) -> ffi::fl_config {
        ffi::fl_config {
            family: self.family as i32,
            qkv_bias: (self.family == Family::Qwen2) as i32, // q/k/v_proj.bias tensors (qwen.rs:93-117)
            hidden_size: self.hidden_size as i64,
            intermediate_size: self.intermediate_size as i64,
            vocab_size: self.vocab_size as i64,
            num_hidden_layers: self.num_hidden_layers as i64,
            num_attention_heads: self.num_attention_heads as i64,
            num_key_value_heads: self.num_key_value_heads.unwrap_or(0) as i64,
            max_position_embeddings: self.max_position_embeddings.unwrap_or(0) as i64,
            sliding_window: self.sliding_window.unwrap_or(0) as i64,
            rms_norm_eps: self.rms_norm_eps,
            rope_theta: self.rope_theta.unwrap_or(0.0),
        }
    }
}

/// One named tensor handed to [`Model::new`]: host bytes (`device < 0`) or a device pointer on HIP device `device`.
/// Borrowed for the duration of the call only.
pub struct TensorView<'a> {
    pub name: &'a str,
    pub dtype: DType,
    pub shape: &'a [usize],
    pub data: *const c_void,
    pub device: i32,
}

impl<'a> TensorView<'a> {
    pub fn host(name: &'a str, dtype: DType, shape: &'a [usize], bytes: &'a [u8]) -> Self {
        TensorView { name, dtype, shape, data: bytes.as_ptr() as *const c_void, device: -1 }
    }
}

/// LogitsProcessor::new(seed, Some(temperature), None) (mod.rs:373-374); `draws_done` = u32 words of the seeded stream
/// already consumed by this request.
#[derive(Clone, Copy, Debug)]
pub struct Sampling {
    pub temperature: f64,
    pub seed: u64,
    pub draws_done: u64,
}
impl Sampling {
    fn to_ffi(self) -> ffi::fl_sampling {
        ffi::fl_sampling { temperature: self.temperature, seed: self.seed, draws_done: self.draws_done }
    }
}

/// `fl_model*`.  Immutable after creation; `Clone` is a reference-count bump (the streaming path clones the model per
/// request, mod.rs:155,181,207).
pub struct Model {
    raw: *mut ffi::fl_model,
    vocab: usize,
}
// SAFETY: the library serialises submission per model and every entry sets its HIP device itself
// (include/fastllm_mi355x.h, "thread-safe: any number of threads ... with DISTINCT caches").
unsafe impl Send for Model {}
unsafe impl Sync for Model {}

impl Clone for Model {
    fn clone(&self) -> Self {
        unsafe { ffi::fl_model_retain(self.raw) };
        Model { raw: self.raw, vocab: self.vocab }
    }
}
impl Drop for Model {
    fn drop(&mut self) {
        unsafe { ffi::fl_model_release(self.raw) };
    }
}

impl Model {
    /// `ModelInitializer::initialize_model` (model_initializer.rs:10-17): single GPU `device_id`.
    pub fn new(cfg: &Config, tensors: &[TensorView<'_>], compute: DType, device_id: i32) -> Result<Model> {
        let names: Vec<CString> = tensors
            .iter()
            .map(|t| CString::new(t.name).map_err(|_| Error { code: ffi::FL_ERR_BAD_ARGUMENT, message: format!("tensor name {:?} contains NUL", t.name) }))
            .collect::<Result<_>>()?;
        let mut descr = Vec::with_capacity(tensors.len());
        for (t, name) in tensors.iter().zip(&names) {
            if t.shape.len() > 4 {
                return Err(Error { code: ffi::FL_ERR_SHAPE_MISMATCH, message: format!("tensor {} has rank {}", t.name, t.shape.len()) });
            }
            let mut shape = [0i64; 4];
            for (d, s) in shape.iter_mut().zip(t.shape) {
                *d = *s as i64;
            }
            descr.push(ffi::fl_tensor { name: name.as_ptr(), dtype: t.dtype as i32, ndim: t.shape.len() as i32, shape, data: t.data, device: t.device, _pad: 0 });
        }
        let c = cfg.to_ffi();
        let ids = [device_id];
        let par = ffi::fl_parallel { mode: ffi::FL_TP_NONE, tp_size: 1, tp_rank: 0, n_device_ids: 1, device_ids: ids.as_ptr(), unique_id: ptr::null() };
        let mut raw: *mut ffi::fl_model = ptr::null_mut();
        // SAFETY: every pointer is valid for the duration of the call; the library copies what it keeps.
        check(unsafe { ffi::fl_model_create(&c, descr.as_ptr(), descr.len(), compute as i32, &par, &mut raw) })?;
        Ok(Model { raw, vocab: cfg.vocab_size })
    }

    pub fn vocab_size(&self) -> usize {
        self.vocab
    }

    pub fn info(&self) -> Result<ffi::fl_model_info> {
        let mut out = ffi::fl_model_info::default();
        check(unsafe { ffi::fl_model_get_info(self.raw, &mut out) })?;
        Ok(out)
    }

    /// `ModelInitializer::initialize_cache` (model_initializer.rs:19; per request, mod.rs:370): caller-owned KV cache.
    pub fn new_cache(&self, max_seq: usize) -> Result<Cache> {
        let mut raw: *mut ffi::fl_cache = ptr::null_mut();
        check(unsafe { ffi::fl_cache_create(self.raw, max_seq, &mut raw) })?;
        Ok(Cache { raw })
    }

    /// `ModelInitializer::forward` (model_initializer.rs:21): last-position logits, `[vocab]` f32.
    pub fn forward(&self, cache: &mut Cache, ids: &[u32], pos: usize) -> Result<Vec<f32>> {
        let mut logits = vec![0f32; self.vocab];
        check(unsafe { ffi::fl_forward(self.raw, cache.raw, ids.as_ptr(), ids.len(), pos, logits.as_mut_ptr()) })?;
        Ok(logits)
    }

    /// The same forward with ArgMax on the device (ties -> last index, Rust `max_by`).
    pub fn forward_argmax(&self, cache: &mut Cache, ids: &[u32], pos: usize) -> Result<u32> {
        let mut tok = 0u32;
        check(unsafe { ffi::fl_forward_argmax(self.raw, cache.raw, ids.as_ptr(), ids.len(), pos, &mut tok) })?;
        Ok(tok)
    }

    /// Seeded temperature sampling on the device (`temperature < 1e-7`: ArgMax, as candle).
    pub fn forward_sample(&self, cache: &mut Cache, ids: &[u32], pos: usize, s: Sampling) -> Result<u32> {
        let mut tok = 0u32;
        let sp = s.to_ffi();
        check(unsafe { ffi::fl_forward_sample(self.raw, cache.raw, ids.as_ptr(), ids.len(), pos, &sp, &mut tok) })?;
        Ok(tok)
    }

    /// The loop body of `Model<M>::generate` (mod.rs:411-453) kept on the device; returns the tokens sampled after each
    /// step, shorter than `n_steps` when `eos` was sampled (which is not included).
    pub fn decode(&self, cache: &mut Cache, first_token: u32, pos: usize, n_steps: usize, eos: Option<u32>, sampling: Option<Sampling>) -> Result<Vec<u32>> {
        let mut toks = vec![0u32; n_steps.max(1)];
        let mut n = 0usize;
        let eos = eos.map(|e| e as i64).unwrap_or(-1);
        let rc = match sampling {
            None => unsafe { ffi::fl_decode_greedy(self.raw, cache.raw, first_token, pos, n_steps, eos, toks.as_mut_ptr(), &mut n) },
            Some(s) => {
                let sp = s.to_ffi();
                unsafe { ffi::fl_decode_sample(self.raw, cache.raw, first_token, pos, n_steps, eos, &sp, toks.as_mut_ptr(), &mut n) }
            }
        };
        check(rc)?;
        toks.truncate(n);
        Ok(toks)
    }

    pub fn synchronize(&self) -> Result<()> {
        check(unsafe { ffi::fl_synchronize(self.raw) })
    }
}

/// `fl_cache*`: the KV cache of one request / stream.  Not `Sync`: one forward at a time per cache.
pub struct Cache {
    raw: *mut ffi::fl_cache,
}
unsafe impl Send for Cache {}
impl Drop for Cache {
    fn drop(&mut self) {
        unsafe { ffi::fl_cache_destroy(self.raw) };
    }
}
impl Cache {
    /// `clear_kv_cache` (mistral.rs:220, qwen.rs:148).
    pub fn reset(&mut self) {
        unsafe { ffi::fl_cache_reset(self.raw) };
    }
    pub fn len(&self) -> usize {
        unsafe { ffi::fl_cache_len(self.raw) }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    pub fn capacity(&self) -> usize {
        unsafe { ffi::fl_cache_capacity(self.raw) }
    }
}

/// `fl_batch*`: up to 64 caches of one model decoded together (one read of the weights per step for all of them).
pub struct Batch<'a> {
    raw: *mut ffi::fl_batch,
    n: usize,
    _caches: std::marker::PhantomData<&'a mut Cache>,
}
impl<'a> Drop for Batch<'a> {
    fn drop(&mut self) {
        unsafe { ffi::fl_batch_destroy(self.raw) };
    }
}
impl<'a> Batch<'a> {
    pub fn new(model: &Model, caches: &'a mut [Cache]) -> Result<Batch<'a>> {
        let raws: Vec<*mut ffi::fl_cache> = caches.iter().map(|c| c.raw).collect();
        let mut raw: *mut ffi::fl_batch = ptr::null_mut();
        check(unsafe { ffi::fl_batch_create(model.raw, raws.as_ptr(), raws.len(), &mut raw) })?;
        Ok(Batch { raw, n: raws.len(), _caches: std::marker::PhantomData })
    }

    /// `n_steps` greedy / sampled steps for every sequence; row `i` holds sequence `i`'s tokens (cut at its EOS).
    pub fn decode(&mut self, first_tokens: &[u32], pos: &[usize], n_steps: usize, eos: Option<u32>, sampling: Option<Sampling>) -> Result<Vec<Vec<u32>>> {
        if first_tokens.len() != self.n || pos.len() != self.n {
            return Err(Error { code: ffi::FL_ERR_BAD_ARGUMENT, message: format!("batch of {} sequences, got {} tokens / {} positions", self.n, first_tokens.len(), pos.len()) });
        }
        let mut toks = vec![0u32; self.n * n_steps.max(1)];
        let mut n_out = vec![0usize; self.n];
        let sp = sampling.map(|s| s.to_ffi());
        let spp = sp.as_ref().map(|s| s as *const ffi::fl_sampling).unwrap_or(ptr::null());
        check(unsafe {
            ffi::fl_batch_decode(self.raw, first_tokens.as_ptr(), pos.as_ptr(), n_steps, eos.map(|e| e as i64).unwrap_or(-1), spp, toks.as_mut_ptr(), n_out.as_mut_ptr())
        })?;
        Ok((0..self.n).map(|i| toks[i * n_steps..i * n_steps + n_out[i]].to_vec()).collect())
    }
}

/// Number of HIP devices the library sees (0 without a GPU: `Model::new` then fails with `FL_ERR_NO_DEVICE`; there is
/// no CPU path).
pub fn device_count() -> usize {
    let mut n = 0;
    let _ = unsafe { ffi::fl_device_count(&mut n) };
    n.max(0) as usize
}

pub fn abi_version() -> i32 {
    unsafe { ffi::fl_abi_version() }
}
