//! Raw bindings of `include/rwkv_abi.h` (ABI version 7), one `pub fn` per export, in the header's order.
//! tests/test_abi_cpu.py diffs this file against the header (names and argument counts) and against the built library.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_void};

pub const RWKV_ABI_VERSION: i32 = 7;

pub type rwkv_status = i32;
pub const RWKV_OK: rwkv_status = 0;
pub const RWKV_ERR_INVALID: rwkv_status = -1;
pub const RWKV_ERR_FORMAT: rwkv_status = -2;
pub const RWKV_ERR_UNSUPPORTED: rwkv_status = -3;
pub const RWKV_ERR_DEVICE: rwkv_status = -4;
pub const RWKV_ERR_OOM: rwkv_status = -5;
pub const RWKV_ERR_NO_STATE: rwkv_status = -6;

pub const RWKV_QUANT_NONE: i32 = 0;
pub const RWKV_QUANT_INT8: i32 = 1;
pub const RWKV_QUANT_NF4: i32 = 2;
pub const RWKV_PRECISION_FP16: i32 = 0;
pub const RWKV_PRECISION_FP32: i32 = 1;
pub const RWKV_PRECISION_FP16_RAW: i32 = 2;
pub const RWKV_ADAPTER_AUTO: i32 = -1;
pub const RWKV_ADAPTER_ECONOMICAL: i32 = -2;
pub const RWKV_OPTION_LAST: i32 = 0;
pub const RWKV_OPTION_FULL: i32 = 1;
pub const RWKV_OPTION_NONE: i32 = 2;
pub const RWKV_SAMPLER_NUCLEUS: i32 = 0;
pub const RWKV_SAMPLER_TYPICAL: i32 = 1;
pub const RWKV_SAMPLER_MIROSTAT: i32 = 2;
pub const RWKV_PROFILE_FAMILIES: usize = 8;

#[repr(C)] pub struct rwkv_engine { _p: [u8; 0] }
#[repr(C)] pub struct rwkv_dstate { _p: [u8; 0] }
#[repr(C)] pub struct rwkv_tokenizer { _p: [u8; 0] }

#[repr(C)] #[derive(Debug, Default, Clone, Copy, PartialEq, Eq)]
pub struct rwkv_model_info { pub version: i32, pub num_layer: i32, pub num_emb: i32, pub num_hidden: i32,
                             pub num_vocab: i32, pub num_head: i32, pub head_size: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rwkv_lora_desc { pub st_bytes: *const u8, pub st_len: usize, pub alpha: c_float }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rwkv_load_desc { pub adapter: i32, pub quant_layers: i32, pub quant_type: i32, pub precision: i32,
                            pub max_batch: i32, pub token_chunk_size: i32,
                            pub st_bytes: *const u8, pub st_len: usize,
                            pub lora: *const rwkv_lora_desc, pub n_lora: usize }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rwkv_slot_input { pub tokens: *const u32, pub n_tokens: usize, pub option: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rwkv_slot_output { pub logits: *mut f32, pub logits_capacity_rows: usize, pub n_rows: usize, pub n_consumed: usize }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rwkv_sample_params { pub top_p: c_float, pub top_k: i32, pub temperature: c_float, pub uniform: c_float,
                                pub adj_tokens: *const u32, pub adj_values: *const c_float, pub n_adj: usize,
                                pub kind: i32, pub tau: c_float, pub allow: *const u8 }

extern "C" {
    // errors / version / adapters  (lib.rs:339-349)
    pub fn rwkv_last_error() -> *const c_char;
    pub fn rwkv_abi_version() -> i32;
    pub fn rwkv_device_count() -> i32;
    pub fn rwkv_device_name(index: i32, buf: *mut c_char, buf_len: usize) -> rwkv_status;
    // Loader::info, load / unload / save  (lib.rs:587, 391-516, 652-656, 131-154)
    pub fn rwkv_model_info_from_st(st_bytes: *const u8, st_len: usize, out: *mut rwkv_model_info) -> rwkv_status;
    pub fn rwkv_engine_create(desc: *const rwkv_load_desc, out: *mut *mut rwkv_engine) -> rwkv_status;
    pub fn rwkv_engine_destroy(e: *mut rwkv_engine);
    pub fn rwkv_engine_save_prefab(e: *mut rwkv_engine, path: *const c_char) -> rwkv_status;
    pub fn rwkv_engine_info(e: *const rwkv_engine, out: *mut rwkv_model_info) -> rwkv_status;
    pub fn rwkv_engine_device(e: *const rwkv_engine) -> i32;
    pub fn rwkv_engine_max_batch(e: *const rwkv_engine) -> i32;
    pub fn rwkv_engine_token_chunk_size(e: *const rwkv_engine) -> i32;
    pub fn rwkv_engine_weight_bytes(e: *const rwkv_engine) -> u64;
    // runtime.infer  (run.rs:1128-1156) and the pinned buffers its logits land in
    pub fn rwkv_infer(e: *mut rwkv_engine, inp: *const rwkv_slot_input, out: *mut rwkv_slot_output) -> rwkv_status;
    pub fn rwkv_host_alloc(bytes: usize, out: *mut *mut c_void) -> rwkv_status;
    pub fn rwkv_host_free(p: *mut c_void);
    pub fn rwkv_plan_chunk(max_batch: i32, token_chunk_size: i32, n_tokens: *const usize, consumed: *mut i32) -> rwkv_status;
    // State  (run.rs:477, 950, 987, 1099-1106) and the /embeddings slice
    pub fn rwkv_state_len(e: *const rwkv_engine) -> usize;
    pub fn rwkv_state_shape(e: *const rwkv_engine, shape: *mut usize);
    pub fn rwkv_state_init(e: *const rwkv_engine, dst: *mut f32) -> rwkv_status;
    pub fn rwkv_state_load(e: *mut rwkv_engine, slot: i32, src: *const f32) -> rwkv_status;
    pub fn rwkv_state_back(e: *mut rwkv_engine, slot: i32, dst: *mut f32) -> rwkv_status;
    pub fn rwkv_state_read(e: *mut rwkv_engine, slot: i32, snap: *mut *mut rwkv_dstate) -> rwkv_status;
    pub fn rwkv_state_write(e: *mut rwkv_engine, slot: i32, snap: *const rwkv_dstate) -> rwkv_status;
    pub fn rwkv_dstate_free(snap: *mut rwkv_dstate);
    pub fn rwkv_state_back_layer(e: *mut rwkv_engine, slot: i32, layer: i32, dst: *mut f32) -> rwkv_status;
    pub fn rwkv_state_back_layer_async(e: *mut rwkv_engine, slot: i32, layer: i32, dst: *mut f32) -> rwkv_status;
    pub fn rwkv_state_sync(e: *mut rwkv_engine) -> rwkv_status;
    pub fn rwkv_read_init_state(e: *const rwkv_engine, st_bytes: *const u8, st_len: usize, dst: *mut f32) -> rwkv_status;
    // softmax task  (run.rs:1178-1183)
    pub fn rwkv_softmax(e: *mut rwkv_engine, inp: *const *const f32, out: *const *mut f32, n_rows: usize) -> rwkv_status;
    // on-device sampling front-end (run.rs:664-697 + sampler/*.rs)
    pub fn rwkv_infer_sample(e: *mut rwkv_engine, inp: *const rwkv_slot_input, sp: *const rwkv_sample_params, out_tokens: *mut u32,
                             out_probs: *mut c_float, emitted: *mut u8, n_consumed: *mut usize) -> rwkv_status;
    // Tokenizer  (lib.rs:375, run.rs:157, 856, bnf.rs:15)
    pub fn rwkv_tokenizer_create(vocab_json: *const c_char, len: usize, out: *mut *mut rwkv_tokenizer) -> rwkv_status;
    pub fn rwkv_tokenizer_destroy(t: *mut rwkv_tokenizer);
    pub fn rwkv_tokenizer_encode(t: *const rwkv_tokenizer, text: *const u8, len: usize, out: *mut u32, cap: usize) -> i64;
    pub fn rwkv_tokenizer_decode(t: *const rwkv_tokenizer, tokens: *const u32, n: usize, out: *mut u8, cap: usize) -> i64;
    pub fn rwkv_tokenizer_token_bytes(t: *const rwkv_tokenizer, token: u32, out: *mut u8, cap: usize) -> i64;
    pub fn rwkv_tokenizer_vocab_size(t: *const rwkv_tokenizer) -> i64;
    // measurement hooks (bench.py / scripts; no reference counterpart)
    pub fn rwkv_profile_family_name(family: i32) -> *const c_char;
    pub fn rwkv_profile_infer(e: *mut rwkv_engine, inp: *const rwkv_slot_input, out: *mut rwkv_slot_output,
                              ms: *mut c_float, launches: *mut i32) -> rwkv_status;
    pub fn rwkv_decode_greedy(e: *mut rwkv_engine, n_slots: i32, first_tokens: *const u32, n_steps: i32,
                              out_tokens: *mut u32, elapsed_ms: *mut c_float) -> rwkv_status;
    pub fn rwkv_bench_gemm(rows: i32, k: i32, fmt: i32, t: i32, hilo: i32, ksw: i32, nmat: i32, iters: i32,
                           us_per_launch: *mut c_float, lds_kib: *mut c_float) -> rwkv_status;
}
