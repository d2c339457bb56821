//! Safe wrapper over `librwkv_hip.so` for ai00-core.  NOT compiled in the image this repository is built in (no rustc there):
//! written against `include/rwkv_abi.h`; `integration/check.sh` type-checks it on a machine with Rust.
//!
//! Layout: the plain wrapper (`Engine`, `Tokenizer`, `Error`) and, in `compat`, the same calls under the names and shapes
//! `crates/ai00-core/src/{lib,run}.rs` import from `web_rwkv` (lib.rs:24-35, run.rs:22-31), so that `ai00-core.patch` is an
//! import change plus the five call sites that construct things.
//!
//! Threading contract (run.rs:1072-1190): per engine one thread calls `infer` + the state functions (the `infer` task) and one
//! calls `softmax` (the `softmax` task).  `Engine` is `Send + Sync` on that contract; the library serialises nothing for you.
use rwkv_hip_sys as sys;
use std::{ffi::{CStr, CString}, os::raw::c_void, ptr, sync::Arc};

#[derive(Debug, thiserror::Error)]
#[error("librwkv_hip error {code}: {message}")]
pub struct Error { pub code: i32, pub message: String }
pub type Result<T> = std::result::Result<T, Error>;

fn check(rc: i32) -> Result<()> {
    if rc == 0 { return Ok(()); }
    let message = unsafe { CStr::from_ptr(sys::rwkv_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

pub use sys::rwkv_model_info as RawInfo;

/// `list_adapters` (lib.rs:339-349)
pub fn list_adapters() -> Vec<String> {
    (0..unsafe { sys::rwkv_device_count() }).filter_map(|i| {
        let mut buf = vec![0i8; 256];
        (unsafe { sys::rwkv_device_name(i, buf.as_mut_ptr() as *mut _, buf.len()) } == 0)
            .then(|| unsafe { CStr::from_ptr(buf.as_ptr() as *const _) }.to_string_lossy().into_owned())
    }).collect()
}

/// `Loader::info(&SafeTensors)` (lib.rs:587).  `bytes`: the mmap of a `.st` file or of a prefab image (sniffed by content).
pub fn model_info(bytes: &[u8]) -> Result<RawInfo> {
    let mut out = RawInfo::default();
    check(unsafe { sys::rwkv_model_info_from_st(bytes.as_ptr(), bytes.len(), &mut out) })?;
    Ok(out)
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum QuantType { None = 0, Int8 = 1, NF4 = 2 }
/// `reload::Precision` (reload.rs:89-94).  `Fp16` holds 1e-3 on logits / state at 32 layers since ABI 7 (the launches that carry a model's
/// f16 operand rounding read hi + lo operands); `Fp16Raw` is the library's all-f16 extension (fastest, not tolerance-holding at depth).
#[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum Precision { Fp16 = 0, Fp32 = 1, Fp16Raw = 2 }
#[derive(Clone, Copy, Debug, PartialEq, Eq)] pub enum Adapter { Auto, Economical, Manual(usize) }
#[derive(Clone, Copy, Debug, PartialEq, Eq, Default)] pub enum OutputOption { #[default] Last = 0, Full = 1, None = 2 }

pub struct LoadDesc<'a> {
    pub adapter: Adapter, pub quant_layers: usize, pub quant_type: QuantType, pub precision: Precision,
    pub max_batch: usize, pub token_chunk_size: usize, pub model: &'a [u8], pub lora: Vec<(&'a [u8], f32)>,
}

/// = Context + Model + vN::Bundle + TokioRuntime<Rnn> + State of the reference (lib.rs:484-516)
#[derive(Debug)]
pub struct Engine { raw: *mut sys::rwkv_engine, pub info: RawInfo, pub max_batch: usize, pub token_chunk_size: usize }
unsafe impl Send for Engine {}
unsafe impl Sync for Engine {}
impl Drop for Engine { fn drop(&mut self) { unsafe { sys::rwkv_engine_destroy(self.raw) } } }

/// device-resident state snapshot (`TensorGpu<f32, ReadWrite>` of run.rs:351-355); clone = share the handle
#[derive(Debug)]
pub struct DeviceState(*mut sys::rwkv_dstate);
unsafe impl Send for DeviceState {}
unsafe impl Sync for DeviceState {}
impl Drop for DeviceState { fn drop(&mut self) { unsafe { sys::rwkv_dstate_free(self.0) } } }

/// pinned host block for the logits of one `infer` call (rwkv_host_alloc): rows land here in one device-to-host copy
#[derive(Debug)]
pub struct PinnedLogits { ptr: *mut f32, floats: usize, in_flight: usize }
unsafe impl Send for PinnedLogits {}
impl PinnedLogits {
    pub fn new(floats: usize) -> Result<Self> {
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { sys::rwkv_host_alloc(floats * 4, &mut p) })?;
        Ok(Self { ptr: p as *mut f32, floats, in_flight: 0 })
    }
    /// Panics while a read-back into the block is in flight: only reachable when a `PendingRows` guard was leaked (`std::mem::forget`) — every
    /// other path holds the block's `&mut` borrow until the copy stream has been waited for.
    pub fn as_slice(&self) -> &[f32] {
        assert!(self.in_flight == 0, "PinnedLogits read while an asynchronous read-back is in flight (a PendingRows guard was leaked)");
        unsafe { std::slice::from_raw_parts(self.ptr, self.floats) }
    }
}
impl Drop for PinnedLogits {
    /// A block with a read-back still accounted in flight (leaked guard) is LEAKED, not freed: the DMA may still be writing it.
    fn drop(&mut self) { if self.in_flight == 0 { unsafe { sys::rwkv_host_free(self.ptr as *mut c_void) } } }
}

/// one slot of an `infer` call: `tokens` is drained by what the call consumed (RnnInputBatch, run.rs:1128)
#[derive(Default, Clone, Debug)]
pub struct SlotInput { pub tokens: Vec<u32>, pub option: OutputOption }

impl Engine {
    pub fn load(d: &LoadDesc) -> Result<Self> {
        let lora: Vec<_> = d.lora.iter().map(|(b, a)| sys::rwkv_lora_desc { st_bytes: b.as_ptr(), st_len: b.len(), alpha: *a }).collect();
        let desc = sys::rwkv_load_desc {
            adapter: match d.adapter { Adapter::Auto => sys::RWKV_ADAPTER_AUTO, Adapter::Economical => sys::RWKV_ADAPTER_ECONOMICAL, Adapter::Manual(n) => n as i32 },
            quant_layers: d.quant_layers as i32, quant_type: d.quant_type as i32, precision: d.precision as i32,
            max_batch: d.max_batch as i32, token_chunk_size: d.token_chunk_size as i32,
            st_bytes: d.model.as_ptr(), st_len: d.model.len(),
            lora: if lora.is_empty() { ptr::null() } else { lora.as_ptr() }, n_lora: lora.len(),
        };
        let mut raw = ptr::null_mut();
        check(unsafe { sys::rwkv_engine_create(&desc, &mut raw) })?;
        let mut info = RawInfo::default();
        check(unsafe { sys::rwkv_engine_info(raw, &mut info) })?;
        let (max_batch, token_chunk_size) = unsafe { (sys::rwkv_engine_max_batch(raw) as usize, sys::rwkv_engine_token_chunk_size(raw) as usize) };
        Ok(Self { raw, info, max_batch, token_chunk_size })
    }
    pub fn save_prefab(&self, path: &str) -> Result<()> {
        let c = CString::new(path).map_err(|_| Error { code: sys::RWKV_ERR_INVALID, message: "path contains NUL".into() })?;
        check(unsafe { sys::rwkv_engine_save_prefab(self.raw, c.as_ptr()) })
    }
    /// Rows each slot can emit in the NEXT call, computed the way the engine computes them: `rwkv_plan_chunk` is the engine's own
    /// split of the `token_chunk_size` budget over the slots (the call consumes at most `token_chunk_size` tokens IN TOTAL, so two
    /// `Full` slots of 100 tokens at chunk 128 emit 128 rows together, not 200); a `Full` slot emits one row per consumed token, a
    /// `Last` slot one row when the call exhausts its tokens, a `None` slot nothing.  The sum is <= token_chunk_size + max_batch.
    pub fn plan_rows(&self, input: &[SlotInput]) -> Result<Vec<usize>> {
        // `rwkv_plan_chunk` reads `max_batch` entries: a shorter slice would be an out-of-bounds read behind a safe signature
        if input.len() != self.max_batch {
            return Err(Error { code: sys::RWKV_ERR_INVALID, message: format!("plan_rows: {} slot inputs for max_batch {}", input.len(), self.max_batch) });
        }
        let pending: Vec<usize> = input.iter().map(|s| s.tokens.len()).collect();
        let mut take = vec![0i32; input.len()];
        check(unsafe { sys::rwkv_plan_chunk(self.max_batch as i32, self.token_chunk_size as i32, pending.as_ptr(), take.as_mut_ptr()) })?;
        Ok(input.iter().zip(&take).map(|(s, &t)| match s.option {
            OutputOption::None => 0,
            OutputOption::Last => usize::from(t > 0 && t as usize == s.tokens.len()),
            OutputOption::Full => t.max(0) as usize }).collect())
    }
    pub fn rows_needed(&self, input: &[SlotInput]) -> Result<usize> { Ok(self.plan_rows(input)?.iter().sum()) }
    /// `runtime.infer(input)` (run.rs:1143): ONE step over <= token_chunk_size tokens.  Consumed tokens are drained from `input`;
    /// returns, per slot, the rows it emitted as `(offset_in_floats, n_rows)` into `logits` (`(0, 0)` for a slot without rows).
    /// Never panics on a caller mistake: a wrong slot count or a block that is too small comes back as `Err` (the infer task treats
    /// it like any other runtime error, run.rs:1143 `?`) instead of taking the server down.
    pub fn infer(&self, input: &mut [SlotInput], logits: &mut PinnedLogits) -> Result<Vec<(usize, usize)>> {
        if input.len() != self.max_batch {
            return Err(Error { code: sys::RWKV_ERR_INVALID, message: format!("infer: {} slot inputs for max_batch {}", input.len(), self.max_batch) });
        }
        let v = self.info.num_vocab as usize;
        let rows = self.plan_rows(input)?;
        let total: usize = rows.iter().sum();
        if total * v > logits.floats {
            return Err(Error { code: sys::RWKV_ERR_INVALID, message: format!("infer: logits block holds {} rows, the step emits {}", logits.floats / v, total) });
        }
        let inp: Vec<_> = input.iter().map(|s| sys::rwkv_slot_input {
            tokens: if s.tokens.is_empty() { ptr::null() } else { s.tokens.as_ptr() }, n_tokens: s.tokens.len(), option: s.option as i32, reserved: 0 }).collect();
        let mut off = 0usize;
        let mut offs = Vec::with_capacity(rows.len());
        let mut out: Vec<_> = rows.iter().map(|&r| {
            let o = sys::rwkv_slot_output { logits: if r == 0 { ptr::null_mut() } else { unsafe { logits.ptr.add(off) } },
                                            logits_capacity_rows: r, n_rows: 0, n_consumed: 0 };
            offs.push(off);
            off += r * v;                         // consecutive pieces of one pinned block: one D2H copy
            o
        }).collect();
        check(unsafe { sys::rwkv_infer(self.raw, inp.as_ptr(), out.as_mut_ptr()) })?;     // fatal for the infer task, like `?` at run.rs:1143
        let mut res = Vec::with_capacity(out.len());
        for ((s, o), &at) in input.iter_mut().zip(&out).zip(&offs) {
            s.tokens.drain(..o.n_consumed);
            res.push(if o.n_rows == 0 { (0, 0) } else { (at, o.n_rows) });      // no pointer arithmetic on the null pointer of a row-less slot
        }
        Ok(res)
    }
    pub fn state_shape(&self) -> [usize; 4] { let mut s = [0usize; 4]; unsafe { sys::rwkv_state_shape(self.raw, s.as_mut_ptr()) }; s }
    pub fn state_init(&self) -> Result<Vec<f32>> {
        let mut v = vec![0f32; unsafe { sys::rwkv_state_len(self.raw) }];
        check(unsafe { sys::rwkv_state_init(self.raw, v.as_mut_ptr()) })?; Ok(v)
    }
    pub fn state_load(&self, slot: usize, data: &[f32]) -> Result<()> {
        if data.len() != unsafe { sys::rwkv_state_len(self.raw) } { return Err(Error { code: sys::RWKV_ERR_INVALID, message: "state tensor has the wrong size".into() }); }
        check(unsafe { sys::rwkv_state_load(self.raw, slot as i32, data.as_ptr()) })
    }
    pub fn state_back(&self, slot: usize) -> Result<Vec<f32>> {
        let mut v = vec![0f32; unsafe { sys::rwkv_state_len(self.raw) }];
        check(unsafe { sys::rwkv_state_back(self.raw, slot as i32, v.as_mut_ptr()) })?; Ok(v)
    }
    pub fn state_read(&self, slot: usize) -> Result<Arc<DeviceState>> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::rwkv_state_read(self.raw, slot as i32, &mut p) })?; Ok(Arc::new(DeviceState(p)))
    }
    pub fn state_write(&self, slot: usize, snap: &DeviceState) -> Result<()> { check(unsafe { sys::rwkv_state_write(self.raw, slot as i32, snap.0) }) }
    /// `/embeddings` (docs/doc-api/openai.md:376-437): one layer's WKV rows `[head_size][num_emb]`
    pub fn state_back_layer(&self, slot: usize, layer: usize) -> Result<Vec<f32>> {
        let mut v = vec![0f32; (self.info.head_size * self.info.num_emb) as usize];
        check(unsafe { sys::rwkv_state_back_layer(self.raw, slot as i32, layer as i32, v.as_mut_ptr()) })?; Ok(v)
    }
    /// The same rows, not waited for: they land in the pinned block at float offset `at` and are valid after the copy stream has been
    /// waited for.  The slot may take its next request at once — a finished document's read-back overlaps the prefill of the following
    /// ones.  Soundness: while a device-to-host copy is in flight the block must be neither read nor freed, and a `&mut` argument alone
    /// does not say that (its borrow ends when the call returns).  So the call hands back a `PendingRows` guard that KEEPS the mutable
    /// borrow of `dst` (and a borrow of the engine): safe code cannot touch, move or drop the block until the guard is consumed by
    /// `PendingRows::sync()` — and dropping the guard without calling it waits too.  Several read-backs into ONE block go through
    /// `PendingRows::also` (disjoint ranges are checked), so the borrow is still held exactly once.
    pub fn state_back_layer_async<'a>(&'a self, slot: usize, layer: usize, dst: &'a mut PinnedLogits, at: usize) -> Result<PendingRows<'a>> {
        let mut p = PendingRows { rt: self, dst, ranges: Vec::new(), waited: false };
        p.issue(slot, layer, at)?;
        Ok(p)
    }
    /// waits for every pending `state_back_layer_async` (what `PendingRows::sync` calls; harmless when nothing is pending)
    pub fn state_sync(&self) -> Result<()> { check(unsafe { sys::rwkv_state_sync(self.raw) }) }
    /// `vN::read_state` (lib.rs:378-389); `Error.code == RWKV_ERR_NO_STATE` <-> the warning at lib.rs:442
    pub fn read_init_state(&self, st: &[u8]) -> Result<Vec<f32>> {
        let mut v = vec![0f32; unsafe { sys::rwkv_state_len(self.raw) }];
        check(unsafe { sys::rwkv_read_init_state(self.raw, st.as_ptr(), st.len(), v.as_mut_ptr()) })?; Ok(v)
    }
    /// `softmax::softmax(&context, Vec<TensorCpu>)` (run.rs:1179), in place; call from the softmax task's thread
    pub fn softmax(&self, rows: &mut [Vec<f32>]) -> Result<()> {
        let inp: Vec<*const f32> = rows.iter().map(|r| r.as_ptr()).collect();
        let out: Vec<*mut f32> = rows.iter_mut().map(|r| r.as_mut_ptr()).collect();
        check(unsafe { sys::rwkv_softmax(self.raw, inp.as_ptr(), out.as_ptr(), rows.len()) })
    }
}

/// Read-backs in flight into one pinned block (`Engine::state_back_layer_async`).  Holds the block's mutable borrow until the copy
/// stream has been waited for: `sync()` gives the block back; `Drop` waits as well, so no path — early return, `?`, panic unwinding —
/// lets the block be read or freed under the DMA.  `std::mem::forget(guard)` is the one safe-code path that ends the borrow without the wait
/// (leaking a guard is safe Rust): `PinnedLogits` therefore counts the read-backs in flight into it (`in_flight`) and its own `Drop` and
/// `as_slice` wait on the engine's copy stream while the count is non-zero.
pub struct PendingRows<'a> {
    rt: &'a Engine,
    dst: &'a mut PinnedLogits,
    ranges: Vec<(usize, usize)>,
    waited: bool,
}
impl<'a> PendingRows<'a> {
    fn issue(&mut self, slot: usize, layer: usize, at: usize) -> Result<()> {
        let n = (self.rt.info.head_size * self.rt.info.num_emb) as usize;
        let end = match at.checked_add(n) { Some(e) if e <= self.dst.floats => e, _ => {
            return Err(Error { code: sys::RWKV_ERR_INVALID, message: format!("pinned block holds {} floats, rows need {}..{}", self.dst.floats, at, at.saturating_add(n)) }) } };
        if self.ranges.iter().any(|&(a, b)| at < b && a < end) {
            return Err(Error { code: sys::RWKV_ERR_INVALID, message: format!("rows {}..{} overlap a read-back already in flight into this block", at, end) });
        }
        check(unsafe { sys::rwkv_state_back_layer_async(self.rt.raw, slot as i32, layer as i32, self.dst.ptr.add(at)) })?;
        self.ranges.push((at, end));
        self.dst.in_flight += 1;
        Ok(())
    }
    /// one more slot's rows into the same block (a disjoint range), still under the one borrow
    pub fn also(&mut self, slot: usize, layer: usize, at: usize) -> Result<()> { self.issue(slot, layer, at) }
    /// wait for the copies; the block is the caller's again
    /// On `Err` the copies may NOT have completed: the guard is dropped at the end of this call with `waited` still false, so its `Drop` waits once
    /// more before the borrow ends, and the block is not handed back.
    pub fn sync(mut self) -> Result<&'a mut PinnedLogits> {
        self.rt.state_sync()?;
        self.waited = true;
        self.dst.in_flight -= self.ranges.len();
        // SAFETY: `self` is consumed and its Drop (below) does nothing once `waited`; the reference is re-borrowed for the original lifetime
        let dst: *mut PinnedLogits = &mut *self.dst;                 // a reborrow, not a move out of a `Drop` type
        Ok(unsafe { &mut *dst })
    }
}
impl Drop for PendingRows<'_> {
    /// waits; only a SUCCESSFUL wait releases the block's in-flight count (after a failed one the block stays unreadable and is leaked on drop)
    fn drop(&mut self) { if !self.waited && self.rt.state_sync().is_ok() { self.dst.in_flight -= self.ranges.len(); } }
}

/// `Tokenizer` (lib.rs:375; run.rs:157-168, 856; sampler/bnf.rs:14-27)
#[derive(Debug)]                       // `RuntimeInfo`, `ThreadRequest` and `CoreRuntime` derive Debug over an `Arc<Tokenizer>` (lib.rs:80, 116; run.rs:374)
pub struct Tokenizer(*mut sys::rwkv_tokenizer);
unsafe impl Send for Tokenizer {}
unsafe impl Sync for Tokenizer {}
impl Drop for Tokenizer { fn drop(&mut self) { unsafe { sys::rwkv_tokenizer_destroy(self.0) } } }
impl Tokenizer {
    pub fn new(vocab_json: &str) -> Result<Self> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::rwkv_tokenizer_create(vocab_json.as_ptr() as *const _, vocab_json.len(), &mut p) })?; Ok(Self(p))
    }
    pub fn encode(&self, text: &[u8]) -> Result<Vec<u32>> {
        let n = unsafe { sys::rwkv_tokenizer_encode(self.0, text.as_ptr(), text.len(), ptr::null_mut(), 0) };
        if n < 0 { return Err(Error { code: n as i32, message: "no matching token found".into() }); }
        let mut v = vec![0u32; n as usize];
        unsafe { sys::rwkv_tokenizer_encode(self.0, text.as_ptr(), text.len(), v.as_mut_ptr(), v.len()) }; Ok(v)
    }
    pub fn decode(&self, tokens: &[u32]) -> Result<Vec<u8>> {
        let n = unsafe { sys::rwkv_tokenizer_decode(self.0, tokens.as_ptr(), tokens.len(), ptr::null_mut(), 0) };
        if n < 0 { return Err(Error { code: n as i32, message: "token index out of range".into() }); }
        let mut v = vec![0u8; n as usize];
        unsafe { sys::rwkv_tokenizer_decode(self.0, tokens.as_ptr(), tokens.len(), v.as_mut_ptr(), v.len()) }; Ok(v)
    }
    /// `token_index_to_bytes` (bnf.rs:15)
    pub fn token_index_to_bytes(&self) -> Vec<Vec<u8>> {
        (0..unsafe { sys::rwkv_tokenizer_vocab_size(self.0) }.max(0) as u32).map(|i| {
            let n = unsafe { sys::rwkv_tokenizer_token_bytes(self.0, i, ptr::null_mut(), 0) };
            let mut v = vec![0u8; n.max(0) as usize];
            if n > 0 { unsafe { sys::rwkv_tokenizer_token_bytes(self.0, i, v.as_mut_ptr(), v.len()) }; }
            v
        }).collect()
    }
}

pub mod compat;
