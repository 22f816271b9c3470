//! MI355X (gfx950) backend for the three causal-LM families: `impl ModelInitializer + ModelArchitecture` on top of
//! libfastllm_mi355x through the `fastllm-mi355x` crate.  Added to the FastLLM tree by
//! `patches/0001-mi355x-backend.patch` (this file is new; the patch only wires it in).
//!
//! Semantics kept from the candle-backed wrappers it replaces:
//!   * Llama rotates by the caller's `pos` (llama.rs:147-149);
//!   * Mistral / Qwen ignore `pos`, rotate by a counter that advances ONCE PER CALL and clear the KV cache when it is 0
//!     (mistral.rs:206-236, qwen.rs:123-151) -- a reference quirk, reproduced so that outputs match it token for token;
//!   * logits come back as `[1, vocab]` f32 on the CPU device, which is what `logits.get(0)?.flatten_all()?` and
//!     `LogitsProcessor::sample` consume (mod.rs:305,421);
//!   * `initialize_cache` has no `&self` (model_initializer.rs:19): the reference hard-codes TinyLlama dimensions for
//!     Llama (llama.rs:125-145).  Here the cache is derived from the model that was loaded last in this process (the
//!     server loads exactly one, main.rs:116-128).
use std::any::Any;
use std::collections::HashMap;
use std::sync::{Mutex, OnceLock};

use anyhow::{anyhow, Context, Result};
use candle_core::{DType, Device, Tensor};
use fastllm_mi355x as mi;

use super::cache::ModelCache;
use super::config::{BaseModelConfig, ModelConfigValidation};
use super::model_initializer::{ModelArchitecture, ModelInitializer};

pub const LLAMA: i32 = 0;
pub const MISTRAL: i32 = 1;
pub const QWEN2: i32 = 2;

/// KV capacity of a per-request cache when config.json gives no smaller bound (positions); FASTLLM_MAX_SEQ overrides.
const DEFAULT_MAX_SEQ: usize = 4096;

fn max_seq_for(cfg_max_pos: usize) -> usize {
    let want = std::env::var("FASTLLM_MAX_SEQ").ok().and_then(|s| s.parse().ok()).unwrap_or(DEFAULT_MAX_SEQ);
    want.min(cfg_max_pos.max(1))
}

/// The model the next `initialize_cache` call belongs to (see the module comment).
static CURRENT: OnceLock<Mutex<Option<(mi::Model, usize)>>> = OnceLock::new();

fn current() -> &'static Mutex<Option<(mi::Model, usize)>> {
    CURRENT.get_or_init(|| Mutex::new(None))
}

#[derive(Clone)]
pub struct Mi355xWithConfig<const FAMILY: i32> {
    model: mi::Model,
}

impl<const FAMILY: i32> std::fmt::Debug for Mi355xWithConfig<FAMILY> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "Mi355xWithConfig<{}>", Self::get_family())
    }
}

/// Caller-owned KV cache plus the per-request call counter the trait asks for (`ModelCache`, cache.rs:5-13: bump, clear,
/// read, downcast).  The counter counts `forward` calls since the last `reset`; the KV handle is cleared with it.
pub struct Mi355xCache {
    kv: mi::Cache,
    calls: usize,
}

impl std::fmt::Debug for Mi355xCache {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("Mi355xCache").field("cached_positions", &self.kv.len()).field("calls", &self.calls).finish()
    }
}

impl ModelCache for Mi355xCache {
    fn increment_offset(&mut self) {
        self.calls = self.calls.saturating_add(1);
    }

    fn reset(&mut self) {
        self.kv.reset();
        self.calls = 0;
    }

    fn get_offset(&self) -> usize {
        self.calls
    }

    fn as_any_mut(&mut self) -> &mut dyn Any {
        self as &mut dyn Any
    }
}

fn family_of(code: i32) -> mi::Family {
    match code {
        LLAMA => mi::Family::Llama,
        MISTRAL => mi::Family::Mistral,
        _ => mi::Family::Qwen2,
    }
}

fn dtype_of(d: DType) -> Result<mi::DType> {
    match d {
        DType::F32 => Ok(mi::DType::F32),
        DType::BF16 => Ok(mi::DType::BF16),
        DType::F16 => Ok(mi::DType::F16),
        other => Err(anyhow!("tensor dtype {:?} is not supported by the MI355X backend (f32, bf16, f16)", other)),
    }
}

/// Host bytes of one candle tensor, in its own dtype (the library casts to the compute dtype while it copies to HBM).
fn tensor_bytes(t: &Tensor) -> Result<(mi::DType, Vec<usize>, Vec<u8>)> {
    let t = t.to_device(&Device::Cpu)?.contiguous()?;
    let shape = t.dims().to_vec();
    let flat = t.flatten_all()?;
    let (dt, bytes) = match t.dtype() {
        DType::F32 => {
            let v: Vec<f32> = flat.to_vec1()?;
            (mi::DType::F32, v.iter().flat_map(|x| x.to_le_bytes()).collect())
        }
        DType::BF16 => {
            let v: Vec<half::bf16> = flat.to_vec1()?;
            (mi::DType::BF16, v.iter().flat_map(|x| x.to_bits().to_le_bytes()).collect())
        }
        DType::F16 => {
            let v: Vec<half::f16> = flat.to_vec1()?;
            (mi::DType::F16, v.iter().flat_map(|x| x.to_bits().to_le_bytes()).collect())
        }
        other => return Err(anyhow!("tensor dtype {:?} is not supported by the MI355X backend", other)),
    };
    Ok((dt, shape, bytes))
}

impl<const FAMILY: i32> ModelInitializer for Mi355xWithConfig<FAMILY> {
    type Config = BaseModelConfig; // config.rs:6-18: exactly the fields fl_config carries
    type Cache = Mi355xCache;

    fn initialize_model(
        config: &Self::Config,
        tensors: HashMap<String, Tensor>,
        dtype: DType,
        _device: &Device,
    ) -> Result<(Self, Self::Cache)> {
        // the reference panics on these (mistral.rs:109-127, qwen.rs:32-37); the library re-checks and returns BAD_CONFIG
        config.validate_head_dimensions()?;
        config.validate_gqa_config()?;
        let cfg = mi::Config {
            family: family_of(FAMILY),
            hidden_size: config.hidden_size,
            intermediate_size: config.intermediate_size,
            vocab_size: config.vocab_size,
            num_hidden_layers: config.num_hidden_layers,
            num_attention_heads: config.num_attention_heads,
            num_key_value_heads: config.num_key_value_heads,
            rms_norm_eps: config.rms_norm_eps,
            rope_theta: config.rope_theta,
            max_position_embeddings: config.max_position_embeddings,
            sliding_window: config.sliding_window,
        };
        // Marshal every named tensor once; the byte buffers must outlive the call, the views borrow them.
        let mut owned: Vec<(String, mi::DType, Vec<usize>, Vec<u8>)> = Vec::with_capacity(tensors.len());
        for (name, t) in tensors.into_iter() {
            let (dt, shape, bytes) = tensor_bytes(&t).with_context(|| format!("marshalling tensor {}", name))?;
            owned.push((name, dt, shape, bytes));
            // `t` is dropped here: the host copy candle made (huggingface.rs:87-88) is released tensor by tensor
        }
        let views: Vec<mi::TensorView<'_>> =
            owned.iter().map(|(n, dt, sh, b)| mi::TensorView::host(n.as_str(), *dt, sh.as_slice(), b.as_slice())).collect();
        let device_id: i32 = std::env::var("FASTLLM_MI355X_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let model = mi::Model::new(&cfg, &views, dtype_of(dtype)?, device_id)
            .map_err(|e| anyhow!("MI355X backend: failed to build the {} model: {}", Self::get_family(), e))?;
        drop(views);
        drop(owned);
        let info = model.info().map_err(|e| anyhow!("{}", e))?;
        let max_seq = max_seq_for(info.cfg.max_position_embeddings as usize);
        *current().lock().unwrap() = Some((model.clone(), max_seq));
        tracing::info!(
            "MI355X backend: {} model resident in HBM ({:.2} GB), KV capacity {} positions per request",
            Self::get_family(),
            info.hbm_bytes_allocated as f64 / 1e9,
            max_seq
        );
        let cache = Mi355xCache { kv: model.new_cache(max_seq).map_err(|e| anyhow!("{}", e))?, calls: 0 };
        Ok((Self { model }, cache))
    }

    fn initialize_cache(_device: &Device, _dtype: DType) -> Result<Self::Cache> {
        let guard = current().lock().unwrap();
        let (model, max_seq) = guard.as_ref().ok_or_else(|| anyhow!("MI355X backend: initialize_cache before initialize_model"))?;
        let kv = model.new_cache(*max_seq).map_err(|e| anyhow!("MI355X backend: KV cache allocation failed: {}", e))?;
        Ok(Mi355xCache { kv, calls: 0 })
    }

    fn forward(&self, input: &Tensor, pos: usize, cache: &mut Self::Cache) -> Result<Tensor> {
        // [1, T] u32 (mod.rs:283-291, 386-394)
        let ids: Vec<u32> = input.flatten_all()?.to_vec1::<u32>()?;
        let rope_pos = if FAMILY == LLAMA {
            pos
        } else {
            if cache.calls == 0 {
                cache.kv.reset(); // clear_kv_cache (mistral.rs:218-221, qwen.rs:146-149)
            }
            cache.calls
        };
        let logits = self
            .model
            .forward(&mut cache.kv, &ids, rope_pos)
            .map_err(|e| anyhow!("MI355X forward failed (T = {}, pos = {}): {}", ids.len(), rope_pos, e))?;
        if FAMILY != LLAMA {
            cache.increment_offset(); // +1 per call, not +T (mistral.rs:234, qwen.rs:143)
        }
        let v = logits.len();
        Ok(Tensor::from_vec(logits, (1, v), &Device::Cpu)?)
    }
}

impl<const FAMILY: i32> ModelArchitecture for Mi355xWithConfig<FAMILY> {
    fn get_family() -> &'static str {
        match FAMILY {
            LLAMA => "Llama",
            MISTRAL => "Mistral",
            _ => "Qwen",
        }
    }

    fn supports_architecture(architecture: &str) -> bool {
        match FAMILY {
            LLAMA => architecture == "LlamaForCausalLM",                       // llama.rs:157-159
            MISTRAL => architecture == "MistralForCausalLM",                   // mistral.rs:244-246
            _ => matches!(architecture, "Qwen2ForCausalLM" | "Qwen2_5_VLForConditionalGeneration"), // qwen.rs:178-183
        }
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn families_and_architectures() {
        assert_eq!(Mi355xWithConfig::<LLAMA>::get_family(), "Llama");
        assert!(Mi355xWithConfig::<MISTRAL>::supports_architecture("MistralForCausalLM"));
        assert!(Mi355xWithConfig::<QWEN2>::supports_architecture("Qwen2_5_VLForConditionalGeneration"));
        assert!(!Mi355xWithConfig::<LLAMA>::supports_architecture("MistralForCausalLM"));
    }

    #[test]
    fn cache_before_model_is_an_error_not_a_panic() {
        // (holds only while no model has been loaded in this test process)
        if current().lock().unwrap().is_none() {
            assert!(Mi355xWithConfig::<LLAMA>::initialize_cache(&Device::Cpu, DType::BF16).is_err());
        }
    }
}
