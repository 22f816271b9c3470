//! Rust binding of `libfastllm_mi355x.so`, the MI355X (gfx950 / CDNA4) decoder forward-pass backend for FastLLM.
//!
//! * [`ffi`]  -- `#[repr(C)]` mirrors of every struct and `extern "C"` declarations of every entry point of
//!   `include/fastllm_mi355x.h` (ABI version 2).  The field offsets are pinned against the C header by
//!   `tests/test_rust_shim.py` (it parses this file, recomputes the C layout and compiles `_Static_assert(offsetof ..)`
//!   lines against the real header), because this repository's build image has no Rust toolchain.
//! * [`safe`] -- RAII handles ([`safe::Model`], [`safe::Cache`], [`safe::Batch`]) and `Result`-returning calls; errors carry
//!   `fl_last_error()`, the analogue of the `anyhow::Error` the reference surfaces (mod.rs:402-405).
//!
//! The reference's plug-in point is the trait `ModelInitializer` (src/models/model_initializer.rs:6-22).  FastLLM is a
//! binary crate, so the trait implementation cannot live here: `patches/0001-mi355x-backend.patch` adds
//! `src/models/mi355x.rs` (`impl ModelInitializer + ModelArchitecture for Mi355xWithConfig<FAMILY>`), the `ModelWrapper`
//! arms (mod.rs:63-70), the registry closures (model_registry.rs:62-109) and the device pick (main.rs:81-97) to the
//! reference tree and makes it depend on this crate.
pub mod ffi;
pub mod safe;

pub use safe::{Batch, Cache, Config, DType, Error, Family, Model, Result, Sampling, TensorView};
