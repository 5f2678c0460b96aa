//! exon-hip: DataFusion glue for the MI355X filter+aggregate path.  NOT COMPILED HERE (no Rust toolchain in the build
//! image); written against datafusion 44 / arrow 53 as pinned by the reference's Cargo.lock.  What CAN be checked without
//! cargo is checked by `tests/test_shim_layout.py`: `sys.rs` is pinned on `include/exon_hip.h` (struct layouts AND every
//! extern signature, type by type), every `use exon::...` path of this crate is checked against the `pub` items of
//! `/root/reference/exon/exon-core/src` (a `pub(crate)` module cannot be imported), every plan shape the rule claims has
//! a matcher, and `tests/abi_harness.c` drives the same call sequence (`open -> push -> finish_arrow -> release`) from C.
//!
//! * [`rule::GpuFilterAggRule`] is the `PhysicalOptimizerRule` that recognises the plans the reference really produces:
//!   `AggregateExec(Partial) <- [CoalesceBatchesExec] <- FilterExec <- {VCFScan | BAMScan | SAMScan | CRAMScan | ...}` for
//!   table scans (exon-core/src/datasources/vcf/scanner.rs:86-162) and `AggregateExec(Partial) <- IndexedVCFScanner` /
//!   `... <- FilterExec <- IndexedVCFScanner | IndexedBAMScan` for pushed-down region filters, which leave NO FilterExec
//!   for the region itself (exon-core/src/datasources/vcf/table_provider.rs:299-320, 571-611) -- and substitutes
//! * [`GpuFilterAggExec`], an `ExecutionPlan` with the SAME output schema as the `AggregateExec(Partial)` it replaces
//!   (group columns, then each aggregate's state fields), so `AggregateExec(Final)` above it is untouched.  Per partition
//!   it either
//!   - hands the partition's files to `exon_hip_scan_open` + `exon_hip_stream_consume_scan` (local files of a `VCFScan` /
//!     `IndexedVCFScanner`, whose `base_config()` is public: bytes go to HBM as they are; inflate, parse, region mask,
//!     filter and aggregate run on the GPU), or
//!   - consumes the child scan's RecordBatches and pushes them through `exon_hip_stream_push` (any object store; BAM / SAM /
//!     CRAM / FASTQ scans, whose file list is private), interning group keys into dictionary ids on the way.
//! * [`udtf::FastqQualityHistogram`] is the table function `fastq_quality_histogram('<path>')` for config 5: DataFusion has
//!   no single operator whose plan could be pattern-matched for a per-position histogram, so the shape gets its own entry
//!   point, registered like the reference's `fastq_scan` (exon-core/src/session_context/exon_context_ext.rs:216-223).
//! * Registration does not touch exon-core: `ExonSession::new(ctx)` accepts any SessionContext and returns `Self`
//!   (exon-core/src/session_context/exon_context_ext.rs:108-112); see INTEGRATION.md section 3.
pub mod rule;
pub mod sys;
pub mod udtf;

use std::any::Any;
use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::sync::Arc;

use arrow::array::{
    Array, ArrayRef, AsArray, Int32Array, ListBuilder, RecordBatch, StringArray, StringBuilder, StructArray, UInt8Array,
};
use arrow::datatypes::{DataType, Field, Schema, SchemaRef};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::common::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_expr::EquivalenceProperties;
use datafusion::physical_plan::execution_plan::{Boundedness, EmissionType};
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, PlanProperties};
use futures::StreamExt;

pub(crate) fn check(ctx: *const sys::exon_hip_ctx, rc: i32) -> Result<()> {
    if rc >= 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::exon_hip_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(DataFusionError::External(format!("exon_hip status {rc}: {msg}").into()))
}

/// Owns the device context + one immutable fused plan; Send + Sync (the C side locks internally: include/exon_hip.h,
/// "ctx / plan are thread-safe").
#[derive(Debug)]
pub struct GpuPlan {
    ctx: *mut sys::exon_hip_ctx,
    plan: *mut sys::exon_hip_plan,
}
unsafe impl Send for GpuPlan {}
unsafe impl Sync for GpuPlan {}
impl GpuPlan {
    pub fn try_new(device: i32, desc: &sys::exon_hip_plan_desc) -> Result<Self> {
        if unsafe { sys::exon_hip_abi_version() } != sys::EXON_HIP_ABI_VERSION {
            return Err(DataFusionError::Plan("libexon_hip.so has a different ABI version than this shim".into()));
        }
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { sys::exon_hip_ctx_create(device, &mut ctx) })?;
        let mut plan = std::ptr::null_mut();
        if let Err(e) = check(ctx, unsafe { sys::exon_hip_plan_create(ctx, desc, &mut plan) }) {
            unsafe { sys::exon_hip_ctx_destroy(ctx) };
            return Err(e);
        }
        Ok(Self { ctx, plan })
    }
}
impl Drop for GpuPlan {
    fn drop(&mut self) {
        unsafe {
            sys::exon_hip_plan_destroy(self.plan);
            sys::exon_hip_ctx_destroy(self.ctx);
        }
    }
}

/// One partition's `exon_hip_stream`.  A stream handle is single-threaded but may MIGRATE between threads between calls
/// (include/exon_hip.h, threading contract) -- exactly what a tokio task does across `.await` -- so the handle is `Send`.
/// Without this newtype the raw pointer held across `input.next().await` makes the future `!Send`, and
/// `SendableRecordBatchStream` (= `Pin<Box<dyn RecordBatchStream + Send>>`) rejects it.
struct StreamHandle(*mut sys::exon_hip_stream);
unsafe impl Send for StreamHandle {}
impl StreamHandle {
    fn open(gpu: &GpuPlan, partition: usize) -> Result<Self> {
        let mut s = std::ptr::null_mut();
        check(gpu.ctx, unsafe { sys::exon_hip_stream_open(gpu.plan, partition as i32, &mut s) })?;
        Ok(Self(s))
    }
    /// moves `batch` into the library: it calls `release` exactly once, success or not -- for a small batch possibly AFTER this
    /// call returns (held until its staging slot is flushed).  `to_ffi` exports heap-owned arrays, as the interface requires.
    fn push(&self, gpu: &GpuPlan, batch: &RecordBatch) -> Result<()> {
        let (mut arr, _sch) = to_ffi(&StructArray::from(batch.clone()).to_data())?; // zero-copy export
        let rc = unsafe { sys::exon_hip_stream_push(self.0, &mut arr as *mut FFI_ArrowArray) };
        std::mem::forget(arr); // released by the library; `_sch` stays ours and is released by its Drop
        check(gpu.ctx, rc)
    }
    /// the stream's group-key dictionary (exon_hip_stream_keys): the value behind every state index, in index order
    fn keys(&self, gpu: &GpuPlan) -> Result<Vec<String>> {
        let (mut n, mut bytes) = (0i32, 0usize);
        check(gpu.ctx, unsafe { sys::exon_hip_stream_keys(self.0, std::ptr::null_mut(), 0, &mut n, &mut bytes, std::ptr::null_mut()) })?;
        let mut buf = vec![0u8; bytes.max(1)];
        check(gpu.ctx, unsafe {
            sys::exon_hip_stream_keys(self.0, buf.as_mut_ptr() as *mut std::os::raw::c_char, bytes, &mut n, &mut bytes, std::ptr::null_mut())
        })?;
        // n '\0'-terminated names back to back ("" = the empty FILTER list is a legal key)
        Ok(buf[..bytes].split(|b| *b == 0).take(n as usize).map(|k| String::from_utf8_lossy(k).into_owned()).collect())
    }
    /// the packed partial state as an Arrow struct array (one row per observed group)
    fn finish(&self, gpu: &GpuPlan) -> Result<StructArray> {
        let mut out = FFI_ArrowArray::empty();
        let mut out_s = FFI_ArrowSchema::empty();
        check(gpu.ctx, unsafe { sys::exon_hip_stream_finish_arrow(self.0, &mut out, &mut out_s) })?;
        let data = unsafe { from_ffi(out, &out_s) }?; // takes ownership of `out`; `out_s` is released by its Drop
        Ok(StructArray::from(data))
    }
}
impl Drop for StreamHandle {
    fn drop(&mut self) {
        unsafe { sys::exon_hip_stream_close(self.0) };
    }
}

/// An open `exon_hip_scan`; only ever used between two `.await`s, closed on drop.
struct ScanHandle(*mut sys::exon_hip_scan);
unsafe impl Send for ScanHandle {}
impl Drop for ScanHandle {
    fn drop(&mut self) {
        unsafe { sys::exon_hip_scan_close(self.0) };
    }
}
/// Which of the fused shapes (BASELINE.json configs 2-5 + the range form) a `GpuFilterAggExec` runs, with what the state
/// batch needs to become the replaced node's output.
#[derive(Debug, Clone)]
pub enum Shape {
    /// `chrom = lit AND pos BETWEEN a AND b` / `vcf_region_filter(...)`, COUNT(*): no group column
    RegionCount,
    /// `reference = lit AND start <= b AND "end" >= a` / `bam_region_filter(...)`, COUNT(*)
    OverlapCount,
    /// `flag & M = V AND CAST(mapping_quality AS INT) >= q`, COUNT(*) GROUP BY reference (Utf8, nullable)
    FlagMapqGroupCount,
    /// `info.F <op> lit`, AVG(qual), COUNT(*) GROUP BY filter (List<Utf8>); `info_field` = F
    CmpAvgByGroup { info_field: String },
    /// per-position histogram of quality_scores (`fastq_quality_histogram`)
    QualPosHist,
}

/// Where the rows of a partition come from.
#[derive(Debug, Clone)]
pub enum Source {
    /// the library opens these local files itself (`exon_hip_scan_open`, gpu_parse = 1); one Vec per partition.
    /// `region`: a pushed-down `vcf_region_filter` / the region a `chrom = .. AND pos ..` conjunction spells, as text.
    Files { format: i32, groups: Vec<Vec<String>>, region: Option<String>, use_index: bool },
    /// the child scan's batches are pushed (`exon_hip_stream_push`); key columns are interned into ids on the way.
    /// `seed_key`: a key that must get id 0 (the literal of `chrom = 'c'` / `reference = 'r'`: the plan compares with id 0).
    ChildBatches { seed_key: Option<String> },
}

#[derive(Debug)]
pub struct GpuFilterAggExec {
    input: Arc<dyn ExecutionPlan>, // the scan (kept as the child so EXPLAIN still shows where rows come from)
    desc: sys::exon_hip_plan_desc, // fused predicate + aggregates
    shape: Shape,
    source: Source,
    partial_schema: SchemaRef, // schema of the replaced AggregateExec(Partial): group columns, then state fields
    props: PlanProperties,
    device: i32,
    gpu: Option<Arc<GpuPlan>>, // always Some since ABI 4 (region shapes name their contig per stream); kept Option for with_new_children
}

/// Arrow types of the partial-aggregate state of a shape, in DataFusion's order (group columns first).
fn state_types(shape: &Shape) -> Vec<DataType> {
    match shape {
        Shape::RegionCount | Shape::OverlapCount => vec![DataType::Int64],
        Shape::FlagMapqGroupCount => vec![DataType::Utf8, DataType::Int64],
        Shape::CmpAvgByGroup { .. } => vec![
            DataType::List(Arc::new(Field::new("item", DataType::Utf8, true))),
            DataType::UInt64,
            DataType::Float64,
            DataType::Int64,
        ],
        Shape::QualPosHist => vec![DataType::Int32, DataType::Int32, DataType::Int64],
    }
}

impl GpuFilterAggExec {
    /// Validates the description against the replaced node's schema and creates the device plan.
    /// `partial_schema` must be `AggregateExec(Partial)::schema()`; its state fields are filled positionally
    /// (count -> Int64, avg -> UInt64 count then Float64 sum), so only the ORDER of DataFusion's state fields matters.
    pub fn try_new(
        input: Arc<dyn ExecutionPlan>,
        desc: sys::exon_hip_plan_desc,
        shape: Shape,
        source: Source,
        partial_schema: SchemaRef,
        device: i32,
    ) -> Result<Self> {
        let want = state_types(&shape);
        let got: Vec<&DataType> = partial_schema.fields().iter().map(|f| f.data_type()).collect();
        if got.len() != want.len() || got.iter().zip(&want).any(|(g, w)| !g.equals_datatype(w)) {
            return Err(DataFusionError::Plan(format!(
                "GpuFilterAggExec: the partial aggregate's schema {got:?} is not the state layout {want:?} of {shape:?}"
            )));
        }
        let n_parts = input.properties().output_partitioning().partition_count();
        if let Source::Files { groups, .. } = &source {
            if groups.len() != n_parts {
                return Err(DataFusionError::Plan("GpuFilterAggExec: one file group per input partition expected".into()));
            }
        }
        // ONE device plan for all partitions.  Region shapes over files name their contig per stream
        // (exon_hip_stream_set_region_contig: every file resolves it in its own header order), so the id in the description
        // is a placeholder; the scan's row mask applies the interval, the plan counts the rows the scan emits.
        let mut desc = desc;
        if matches!((&shape, &source), (Shape::RegionCount | Shape::OverlapCount, Source::Files { .. })) {
            desc.region_chrom_id = 0;
            desc.region_start = 1;
            desc.region_end = sys::EXON_HIP_REGION_OPEN_END;
        }
        // fails at plan time when there is no GPU or the library is missing: the rule then keeps DataFusion's plan
        let gpu = Some(Arc::new(GpuPlan::try_new(device, &desc)?));
        // same partitioning as the scan (a RepartitionExec(RoundRobinBatch) the rule peeled off is gone with the nodes it
        // fed); one state batch per partition, emitted at the end
        let props = PlanProperties::new(
            EquivalenceProperties::new(partial_schema.clone()),
            input.properties().output_partitioning().clone(),
            EmissionType::Final,
            Boundedness::Bounded,
        );
        Ok(Self { input, desc, shape, source, partial_schema, props, device, gpu })
    }
}

impl DisplayAs for GpuFilterAggExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        let src = match &self.source {
            Source::Files { region: Some(r), use_index, .. } => format!("files->HBM, region={r}, indexed={use_index}"),
            Source::Files { .. } => "files->HBM".to_string(),
            Source::ChildBatches { .. } => "child batches".to_string(),
        };
        write!(f, "GpuFilterAggExec: kind={}, shape={:?}, source={}", self.desc.kind, self.shape, src)
    }
}

/// group id -> key strings of this partition (ids are dictionary ids of the scan or of the interner below)
struct Keys(Vec<Option<String>>);

/// The state batch of `exon_hip_stream_finish_arrow` -> the replaced node's output batch: group ids become the original key
/// type (Utf8 reference / List<Utf8> filter), state columns keep their order.
fn state_to_partial(shape: &Shape, state: &StructArray, keys: &Keys, schema: &SchemaRef) -> Result<RecordBatch> {
    let cols: Vec<ArrayRef> = match shape {
        Shape::RegionCount | Shape::OverlapCount => vec![state.column(0).clone()],
        Shape::FlagMapqGroupCount => {
            let ids = state.column(0).as_primitive::<arrow::datatypes::Int32Type>();
            let names: StringArray =
                ids.iter().map(|g| g.and_then(|g| keys.0.get(g as usize).cloned().flatten())).collect();
            vec![Arc::new(names), state.column(1).clone()]
        }
        Shape::CmpAvgByGroup { .. } => {
            let ids = state.column(0).as_primitive::<arrow::datatypes::Int32Type>();
            let mut b = ListBuilder::new(StringBuilder::new());
            for g in ids.iter() {
                // a FILTER list travels as its ';'-joined text, "" = the empty list (never NULL: lazy_array_builder.rs:209-216)
                let text = g.and_then(|g| keys.0.get(g as usize).cloned().flatten()).unwrap_or_default();
                for item in text.split(';').filter(|s| !s.is_empty()) {
                    b.values().append_value(item);
                }
                b.append(true);
            }
            vec![Arc::new(b.finish()), state.column(1).clone(), state.column(2).clone(), state.column(3).clone()]
        }
        Shape::QualPosHist => vec![state.column(0).clone(), state.column(1).clone(), state.column(2).clone()],
    };
    RecordBatch::try_new(schema.clone(), cols).map_err(DataFusionError::from)
}

/// Rewrites the key column of a child batch into int32 ids (order of first appearance within the partition) so the batch
/// has the device layout `exon_hip_stream_push` expects.  Only the ChildBatches source pays for this.
fn intern_keys(batch: &RecordBatch, key_col: usize, ids: &mut HashMap<Option<String>, i32>, keys: &mut Keys) -> Result<RecordBatch> {
    let col = batch.column(key_col);
    let mut out = Vec::with_capacity(col.len());
    let mut valid = Vec::with_capacity(col.len());
    for row in 0..col.len() {
        let k: Option<String> = match col.data_type() {
            DataType::Utf8 => {
                let a = col.as_string::<i32>();
                (!a.is_null(row)).then(|| a.value(row).to_string())
            }
            DataType::List(_) => {
                let items = col.as_list::<i32>().value(row);
                let s = items.as_string::<i32>();
                Some((0..s.len()).map(|i| s.value(i)).collect::<Vec<_>>().join(";"))
            }
            other => return Err(DataFusionError::Plan(format!("cannot intern a key column of type {other}"))),
        };
        valid.push(k.is_some());
        let next = ids.len() as i32;
        let id = *ids.entry(k.clone()).or_insert_with(|| {
            keys.0.push(k);
            next
        });
        out.push(id);
    }
    // a NULL key (unmapped read's reference) stays NULL in the id column: K3 gives NULL its own group, K2 / K6 drop the row
    let id_col = Int32Array::from(out.into_iter().zip(valid).map(|(v, ok)| ok.then_some(v)).collect::<Vec<Option<i32>>>());
    let mut cols = batch.columns().to_vec();
    cols[key_col] = Arc::new(id_col);
    let mut fields: Vec<Field> = batch.schema().fields().iter().map(|f| f.as_ref().clone()).collect();
    fields[key_col] = Field::new(fields[key_col].name(), DataType::Int32, true);
    RecordBatch::try_new(Arc::new(Schema::new(fields)), cols).map_err(DataFusionError::from)
}

/// `mapping_quality` is Utf8 in the reference's SAM/BAM schema (exon-sam/src/schema_builder.rs:392: decimal text, NULL for
/// 255, exon-bam/src/array_builder.rs:136-143); the device layout is u8 + validity.  `CAST(mapping_quality AS INT)` of a
/// non-numeric string is an error in DataFusion; here such a value becomes NULL (the row is dropped either way).
fn mapq_to_u8(batch: &RecordBatch, col: usize) -> Result<RecordBatch> {
    let c = batch.column(col);
    let a = match c.data_type() {
        DataType::Utf8 => c.as_string::<i32>(),
        DataType::UInt8 => return Ok(batch.clone()),
        other => return Err(DataFusionError::Plan(format!("mapping_quality: expected Utf8, found {other}"))),
    };
    let out: UInt8Array = (0..a.len())
        .map(|i| if a.is_null(i) { None } else { a.value(i).trim().parse::<u8>().ok() })
        .collect();
    let mut cols = batch.columns().to_vec();
    cols[col] = Arc::new(out);
    let mut fields: Vec<Field> = batch.schema().fields().iter().map(|f| f.as_ref().clone()).collect();
    fields[col] = Field::new(fields[col].name(), DataType::UInt8, true);
    RecordBatch::try_new(Arc::new(Schema::new(fields)), cols).map_err(DataFusionError::from)
}

/// name part of a region string ("chr1:1-100" -> "chr1")
fn region_name(region: &str) -> &str {
    region.split(':').next().unwrap_or(region)
}

impl ExecutionPlan for GpuFilterAggExec {
    fn name(&self) -> &str {
        "GpuFilterAggExec"
    }
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn properties(&self) -> &PlanProperties {
        &self.props
    }
    fn schema(&self) -> SchemaRef {
        self.partial_schema.clone()
    }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> {
        vec![&self.input]
    }
    fn with_new_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(Arc::new(Self {
            input: c[0].clone(),
            desc: self.desc,
            shape: self.shape.clone(),
            source: self.source.clone(),
            partial_schema: self.partial_schema.clone(),
            props: self.props.clone(),
            device: self.device,
            gpu: self.gpu.clone(),
        }))
    }

    /// One HIP stream per partition; partitions = file groups (regroup_files_by_size), one GPU each in a multi-GPU job.
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let gpu = self.gpu.clone();
        let device = self.device;
        let desc = self.desc;
        let shape = self.shape.clone();
        let schema = self.partial_schema.clone();
        let out_schema = schema.clone();
        let source = self.source.clone();
        // operator argument order (include/exon_hip.h, exon_hip_plan_desc.columns): K2 / K6 key = argument 0,
        // K3 / K4 group key = argument 2, K3 mapq = argument 1
        let key_arg = match shape {
            Shape::RegionCount | Shape::OverlapCount => 0,
            _ => 2,
        };
        let key_col = self.desc.columns[key_arg] as usize;
        let mapq_col = self.desc.columns[1] as usize;
        let mut input = match &source {
            Source::ChildBatches { .. } => Some(self.input.execute(partition, ctx)?),
            Source::Files { .. } => None,
        };
        let fut = async move {
            let mut out: Vec<RecordBatch> = Vec::new();
            match &source {
                Source::Files { format, groups, region, use_index } => {
                    // Whole files are decoded and aggregated inside ONE blocking call per file (tens of milliseconds to
                    // seconds): that must not run on a tokio worker, where it would stall every other task of the runtime
                    // (the reference's scans are async all the way down: exon-core/src/datasources/vcf/scanner.rs:142-162).
                    // The handles are Send (see StreamHandle / ScanHandle), the closure owns clones of everything it needs.
                    let (format, use_index) = (*format, *use_index);
                    let files = groups[partition].clone();
                    let region = region.clone();
                    let (shape_b, schema_b) = (shape.clone(), schema.clone());
                    let plan = gpu.clone().expect("GpuFilterAggExec::try_new always creates the plan");
                    let batches = tokio::task::spawn_blocking(move || -> Result<Vec<RecordBatch>> {
                        let info = match &shape_b {
                            Shape::CmpAvgByGroup { info_field } => Some(CString::new(info_field.as_str()).unwrap()),
                            _ => None,
                        };
                        let region_c = region.as_ref().map(|r| CString::new(r.as_str()).unwrap());
                        // ONE stream per partition: group keys travel by VALUE (ABI 4) -- the first file's dictionary
                        // becomes the stream's, every further file is aggregated under its own ids and added in under
                        // the stream's -- so a partition emits one state batch however many files it holds
                        let open_stream = || -> Result<StreamHandle> {
                            let stream = StreamHandle::open(&plan, partition)?;
                            if matches!(shape_b, Shape::RegionCount | Shape::OverlapCount) {
                                // `chrom = <id>` is resolved by NAME in every file's own header order; the scan's row mask
                                // (k_region_mask) has already applied the interval, so the plan counts what the scan emits
                                let name = CString::new(region_name(region.as_deref().unwrap_or(""))).unwrap();
                                check(plan.ctx, unsafe { sys::exon_hip_stream_set_region_contig(stream.0, name.as_ptr()) })?;
                            }
                            Ok(stream)
                        };
                        // the state of one stream as the partial-aggregate batch AggregateExec(Final) consumes
                        let emit = |stream: &StreamHandle| -> Result<RecordBatch> {
                            // key VALUES of the state's indexes (VCF: FILTER lists as ';'-joined text; BAM: reference names)
                            let keys = Keys(stream.keys(&plan)?.into_iter().map(Some).collect());
                            let state = stream.finish(&plan)?;
                            state_to_partial(&shape_b, &state, &keys, &schema_b)
                        };
                        let mut done: Vec<RecordBatch> = Vec::new();
                        let mut stream = open_stream()?;
                        for path in &files {
                            let opt = sys::exon_hip_scan_options {
                                format,
                                compression: sys::EXON_HIP_COMPRESSION_AUTO,
                                batch_size: 0,
                                info_field: info.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()),
                                region: region_c.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()),
                                use_index: use_index as i32,
                                gpu_parse: 1,
                                projection: 0, // the fused kernels read chrom / pos / qual / filter / typed INFO only
                            };
                            let cpath = CString::new(path.as_str()).unwrap();
                            let mut raw = std::ptr::null_mut();
                            check(std::ptr::null(), unsafe { sys::exon_hip_scan_open(cpath.as_ptr(), &opt, &mut raw) })?;
                            let scan = ScanHandle(raw);
                            let mut rows = 0i64;
                            let mut rc = unsafe { sys::exon_hip_stream_consume_scan(stream.0, scan.0, &mut rows) };
                            if rc == sys::EXON_HIP_ECAPACITY {
                                // The UNION of the dictionaries of this partition's files no longer fits the plan's n_groups
                                // (each file's own does, or the retry below fails too).  The library left the stream as it
                                // was before this file: emit its state as one partial batch -- AggregateExec(Final) merges
                                // partial batches by key VALUE, any number of them per partition -- and give the file a
                                // fresh stream (what every file had before the partition shared one).
                                // (a stream that has consumed nothing yet -- the failing file was its first -- is
                                // empty after ECAPACITY: no keys, no rows, nothing to emit)
                                if !stream.keys(&plan)?.is_empty() {
                                    done.push(emit(&stream)?);
                                }
                                stream = open_stream()?;
                                drop(scan);
                                let mut raw2 = std::ptr::null_mut();
                                check(std::ptr::null(), unsafe { sys::exon_hip_scan_open(cpath.as_ptr(), &opt, &mut raw2) })?;
                                let scan2 = ScanHandle(raw2);
                                rc = unsafe { sys::exon_hip_stream_consume_scan(stream.0, scan2.0, &mut rows) };
                            }
                            check(plan.ctx, rc)?;
                        }
                        done.push(emit(&stream)?);
                        Ok(done)
                    })
                    .await
                    .map_err(|e| DataFusionError::External(Box::new(e)))??;
                    out.extend(batches);
                }
                Source::ChildBatches { seed_key } => {
                    let plan = gpu.as_ref().expect("ChildBatches always has a shared plan").clone();
                    let stream = StreamHandle::open(&plan, partition)?; // Send: may be held across the awaits below
                    let mut keys = Keys(Vec::new());
                    let mut ids: HashMap<Option<String>, i32> = HashMap::new();
                    if let Some(k) = seed_key {
                        ids.insert(Some(k.clone()), 0); // the plan compares the key column with id 0
                        keys.0.push(Some(k.clone()));
                    }
                    let input = input.as_mut().unwrap();
                    while let Some(batch) = input.next().await {
                        let mut batch: RecordBatch = batch?;
                        if !matches!(shape, Shape::QualPosHist) {
                            batch = intern_keys(&batch, key_col, &mut ids, &mut keys)?;
                        }
                        if matches!(shape, Shape::FlagMapqGroupCount) {
                            batch = mapq_to_u8(&batch, mapq_col)?;
                        }
                        stream.push(&plan, &batch)?;
                    }
                    let state = stream.finish(&plan)?;
                    out.push(state_to_partial(&shape, &state, &keys, &schema)?);
                }
            }
            Ok::<_, DataFusionError>(futures::stream::iter(out.into_iter().map(Ok::<RecordBatch, DataFusionError>)))
        };
        let batches = futures::stream::once(fut).map(|r| match r {
            Ok(s) => s.boxed(),
            Err(e) => futures::stream::once(async move { Err(e) }).boxed(),
        });
        Ok(Box::pin(RecordBatchStreamAdapter::new(out_schema, batches.flatten())))
    }
}
