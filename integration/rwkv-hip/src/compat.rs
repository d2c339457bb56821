//! The `web_rwkv` items ai00-core imports (lib.rs:24-35, run.rs:22-31), re-expressed over `Engine`, with the method
//! names and shapes the two files use — enough for `ai00-core.patch` to be an import change plus the construction sites.
//! What is NOT here: `wgpu` adapters (replaced by `list_adapters`), `ContextBuilder::auto_limits`, `v4` (unsupported:
//! RWKV_ERR_UNSUPPORTED), the CBOR `Seed`/`Prefab` (the prefab image is this library's own; `Engine::load` sniffs it).
use crate::{Adapter, DeviceState, Engine, Error, LoadDesc, OutputOption, PinnedLogits, Precision, QuantType, SlotInput};
use serde::{Deserialize, Serialize};
use std::sync::{Arc, Mutex};

pub type TensorError = Error;
pub type RuntimeError = Error;

// serde: `RuntimeInfo` / `FileInfo` / `InfoResponse` carry a `ModelInfo` to the HTTP layer (ai00-server api/model.rs:15, api/file.rs:68)
#[derive(Debug, Clone, Copy, PartialEq, Eq, Serialize, Deserialize)] pub enum ModelVersion { V4, V5, V6, V7 }
#[derive(Debug, Clone, Serialize, Deserialize)]
pub struct ModelInfo { pub version: ModelVersion, pub num_layer: usize, pub num_emb: usize, pub num_hidden: usize,
                       pub num_vocab: usize, pub num_head: usize }
impl From<crate::RawInfo> for ModelInfo {
    fn from(i: crate::RawInfo) -> Self {
        let version = match i.version { 5 => ModelVersion::V5, 6 => ModelVersion::V6, _ => ModelVersion::V7 };
        Self { version, num_layer: i.num_layer as usize, num_emb: i.num_emb as usize, num_hidden: i.num_hidden as usize,
               num_vocab: i.num_vocab as usize, num_head: i.num_head as usize }
    }
}
pub struct Loader;
impl Loader { pub fn info(bytes: &[u8]) -> Result<ModelInfo, Error> { crate::model_info(bytes).map(Into::into) } }

/// `reload::Model::quant_type` / `ReloadRequest::quant_type` (reload.rs:27, lib.rs:207): read from `Config.toml` and the `/api/models/load` body
#[derive(Debug, Default, Clone, Copy, PartialEq, Eq, Hash, Serialize, Deserialize)] pub enum Quant { #[default] None, Int8, NF4, SF4 }

/// `TensorCpu<f32>`: shape `[x, y, z, w]` (x fastest) + data, the only form ai00-core uses (run.rs:131, 202, 314, 666-697, 987)
/// serde (`Arc` needs serde's `rc` feature, Cargo.toml): `InitState` derives both (lib.rs:294-301; cbor `.state` files, run.rs:406)
#[derive(Debug, Clone, Serialize, Deserialize)]
pub struct TensorCpu<T> { shape: [usize; 4], data: Arc<Vec<T>> }
impl<T: Clone> TensorCpu<T> {
    pub fn from_data(shape: [usize; 4], data: Vec<T>) -> Result<Self, Error> {
        if shape.iter().product::<usize>() != data.len() { return Err(Error { code: -1, message: "tensor size mismatch".into() }); }
        Ok(Self { shape, data: Arc::new(data) })
    }
    pub fn shape(&self) -> [usize; 4] { self.shape }
    pub fn to_vec(&self) -> Vec<T> { self.data.as_ref().clone() }
    pub fn map<U>(&self, f: impl FnMut(&T) -> U) -> TensorCpu<U> { TensorCpu { shape: self.shape, data: Arc::new(self.data.iter().map(f).collect()) } }
    pub fn len(&self) -> usize { self.data.len() }
    pub fn is_empty(&self) -> bool { self.data.is_empty() }
    /// rows of a `[V, n, 1, 1]` output (run.rs:735-741 walks them)
    pub fn split(&self, axis: usize) -> Result<Vec<TensorCpu<T>>, Error> {
        assert_eq!(axis, 1);
        let v = self.shape[0];
        Ok(self.data.chunks(v).map(|c| TensorCpu { shape: [v, 1, 1, 1], data: Arc::new(c.to_vec()) }).collect())
    }
}
impl<T> std::ops::Deref for TensorCpu<T> { type Target = [T]; fn deref(&self) -> &[T] { &self.data } }
/// `TensorGpu<f32, ReadWrite>` as ai00-core uses it: an opaque state snapshot (run.rs:351-355, 772-789)
#[derive(Debug, Clone)] pub struct TensorGpu(pub Arc<DeviceState>);   // Debug: `InferBatch` derives it (run.rs:327)

#[derive(Debug, Clone, Copy, PartialEq, Eq, Default)] pub enum RnnOption { #[default] Last, Full }
#[derive(Debug, Clone, Default)] pub struct RnnInputBatch { pub tokens: Vec<u32>, pub option: RnnOption }
impl RnnInputBatch { pub fn new(tokens: Vec<u32>, option: RnnOption) -> Self { Self { tokens, option } } }
#[derive(Debug, Clone)] pub struct RnnInput { pub batches: Vec<RnnInputBatch>, pub token_chunk_size: usize }
impl RnnInput {
    pub fn new(batches: Vec<RnnInputBatch>, token_chunk_size: usize) -> Self { Self { batches, token_chunk_size } }
    pub fn num_token(&self) -> usize { self.batches.iter().map(|b| b.tokens.len()).sum() }
}
#[derive(Debug, Clone)] pub struct RnnOutputBatch(pub TensorCpu<f32>);
impl RnnOutputBatch { pub fn is_empty(&self) -> bool { self.0.is_empty() } }

/// `Arc<dyn Runtime<Rnn>>` + `Arc<dyn State>` + the `Context` the softmax task holds: all views of one engine
#[derive(Debug, Clone)]
pub struct Runtime { engine: Arc<Engine>, logits: Arc<Mutex<PinnedLogits>> }
#[derive(Debug, Clone)] pub struct State { engine: Arc<Engine> }
#[derive(Debug, Clone)] pub struct Context { engine: Arc<Engine> }   // Debug: `CoreRuntime` derives it with a `Context` field (run.rs:365-369)

impl Runtime {
    /// `ModelSerialize::serialize` (lib.rs:131-154): the loaded model as one prefab image
    pub fn save_prefab(&self, path: &str) -> Result<(), Error> { self.engine.save_prefab(path) }
    /// `runtime.infer(input).await` (run.rs:1143).  Blocking FFI: call it on the dedicated infer task thread / spawn_blocking.
    pub fn infer(&self, input: RnnInput) -> Result<(RnnInput, Vec<RnnOutputBatch>), Error> {
        let mut slots: Vec<SlotInput> = input.batches.iter().map(|b| SlotInput {
            tokens: b.tokens.clone(), option: if b.option == RnnOption::Full { OutputOption::Full } else { OutputOption::Last } }).collect();
        let v = self.engine.info.num_vocab as usize;
        let mut block = self.logits.lock().unwrap();
        let rows = self.engine.infer(&mut slots, &mut block)?;
        let out = rows.iter().map(|&(off, n)| RnnOutputBatch(TensorCpu { shape: [v, n, 1, 1], data: Arc::new(block.as_slice()[off..off + n * v].to_vec()) })).collect();
        let batches = slots.into_iter().zip(input.batches).map(|(s, b)| RnnInputBatch { tokens: s.tokens, option: b.option }).collect();
        Ok((RnnInput { batches, token_chunk_size: input.token_chunk_size }, out))
    }
}
impl State {
    pub fn init(&self) -> TensorCpu<f32> { let s = self.engine.state_shape(); TensorCpu { shape: s, data: Arc::new(self.engine.state_init().expect("state_init")) } }
    pub fn load(&self, tensor: TensorCpu<f32>, batch: usize) -> Result<(), Error> { self.engine.state_load(batch, &tensor) }
    pub fn back(&self, batch: usize) -> Result<TensorCpu<f32>, Error> { Ok(TensorCpu { shape: self.engine.state_shape(), data: Arc::new(self.engine.state_back(batch)?) }) }
    pub fn read(&self, batch: usize) -> Result<TensorGpu, Error> { self.engine.state_read(batch).map(TensorGpu) }
    pub fn write(&self, tensor: TensorGpu, batch: usize) -> Result<(), Error> { self.engine.state_write(batch, &tensor.0) }
    /// `vN::read_state(context, info, model)` (lib.rs:378-389) on the bytes of a `.state` / state-tuned model file
    pub fn read_init_state(&self, st: &[u8]) -> Result<TensorCpu<f32>, Error> {
        Ok(TensorCpu { shape: self.engine.state_shape(), data: Arc::new(self.engine.read_init_state(st)?) })
    }
    /// `/embeddings`: one layer's rows instead of the whole slab
    pub fn embed(&self, layer: usize, batch: usize) -> Result<TensorCpu<f32>, Error> {
        let (n, c) = (self.engine.info.head_size as usize, self.engine.info.num_emb as usize);
        Ok(TensorCpu { shape: [c, n, 1, 1], data: Arc::new(self.engine.state_back_layer(batch, layer)?) })
    }
}
impl Context {
    pub fn tensor_from_data(&self, shape: [usize; 4], data: Vec<f32>) -> Result<TensorCpu<f32>, Error> { TensorCpu::from_data(shape, data) }
}
/// `web_rwkv::runtime::softmax::softmax(&context, Vec<TensorCpu>)` (run.rs:1179)
pub fn softmax(context: &Context, input: Vec<TensorCpu<f32>>) -> Result<Vec<TensorCpu<f32>>, Error> {
    let mut rows: Vec<Vec<f32>> = input.iter().map(|t| t.to_vec()).collect();
    context.engine.softmax(&mut rows)?;
    Ok(rows.into_iter().zip(input).map(|(d, t)| TensorCpu { shape: t.shape, data: Arc::new(d) }).collect())
}

/// `ModelBuilder::new(ctx, st).quant(map).lora(l).build_vN()` + `Bundle::new(model, max_batch)` + `TokioRuntime::new(bundle)`
pub struct ModelBuilder<'a> { adapter: Adapter, model: &'a [u8], quant: (usize, QuantType), lora: Vec<(&'a [u8], f32)> }
impl<'a> ModelBuilder<'a> {
    pub fn new(adapter: Adapter, model: &'a [u8]) -> Self { Self { adapter, model, quant: (0, QuantType::None), lora: vec![] } }
    /// lib.rs:465: layers `0..quant` get `quant_type`
    pub fn quant(mut self, layers: usize, quant: Quant) -> Result<Self, Error> {
        let q = match quant { Quant::None => QuantType::None, Quant::Int8 => QuantType::Int8, Quant::NF4 => QuantType::NF4,
                              Quant::SF4 => return Err(Error { code: -3, message: "SF4 is not supported by this backend".into() }) };
        self.quant = (layers, q); Ok(self)
    }
    /// `LoraBlend::full(alpha)` (lib.rs:466-482)
    pub fn lora(mut self, data: &'a [u8], alpha: f32) -> Self { self.lora.push((data, alpha)); self }
    /// `fp32` = `matches!(precision, Precision::Fp32)` (lib.rs:503-515).  `false` is the reference's default `Precision::Fp16`, which in this
    /// backend is the tolerance-holding fp16 mode (RWKV_PRECISION_FP16, ABI 7): nothing for ai00-core to select — the all-f16 raw mode is only
    /// reachable through `rwkv_hip::LoadDesc { precision: Precision::Fp16Raw, .. }`.
    pub fn build(self, max_batch: usize, token_chunk_size: usize, fp32: bool) -> Result<(Runtime, State, Context, ModelInfo), Error> {
        let engine = Arc::new(Engine::load(&LoadDesc { adapter: self.adapter, quant_layers: self.quant.0, quant_type: self.quant.1,
            precision: if fp32 { Precision::Fp32 } else { Precision::Fp16 }, max_batch, token_chunk_size, model: self.model, lora: self.lora })?);
        let info: ModelInfo = engine.info.into();
        let logits = PinnedLogits::new((token_chunk_size + max_batch) * info.num_vocab)?;
        Ok((Runtime { engine: engine.clone(), logits: Arc::new(Mutex::new(logits)) }, State { engine: engine.clone() }, Context { engine }, info))
    }
}
pub use crate::Tokenizer;
