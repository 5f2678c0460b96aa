//! exon-hip: DataFusion glue for the MI355X filter+aggregate path.  NOT COMPILED HERE (no Rust toolchain in the build
//! image); written against datafusion 44 / arrow 53 as pinned by the reference's Cargo.lock.  What CAN be checked without
//! cargo is checked: `tests/test_shim_layout.py` pins `sys.rs` on `include/exon_hip.h`, and `tests/abi_harness.c` drives the
//! same call sequence (`open -> push (hand-built ArrowArray) -> finish_arrow -> release`) from plain C.
//!
//! * [`rule::GpuFilterAggRule`] is the `PhysicalOptimizerRule` that recognises
//!   `AggregateExec(Partial) <- [CoalesceBatchesExec] <- FilterExec <- {VCFScan | BAMScan | FASTQScan}` (the plan shape
//!   of exon-core/src/datasources/vcf/table_provider.rs:571-611) for the four fused query shapes and substitutes
//! * [`GpuFilterAggExec`], an `ExecutionPlan` with the SAME output schema as the `AggregateExec(Partial)` it replaces (group
//!   columns, then each aggregate's state fields), so `AggregateExec(Final)` above it is untouched.  Per partition it either
//!   - hands the partition's files to `exon_hip_scan_open` + `exon_hip_stream_consume_scan` (local files of a `VCFScan`,
//!     whose `base_config()` is public: bytes go to HBM as they are, inflate + parse + filter + aggregate on the GPU), or
//!   - consumes the child scan's RecordBatches and pushes them through `exon_hip_stream_push` (any object store, BAM /
//!     FASTQ scans whose file list is private), interning group keys into dictionary ids on the way.
//! * Registration does not touch exon-core: `ExonSession::new(ctx)` accepts any SessionContext
//!   (exon-core/src/session_context/exon_context_ext.rs:103-112); see INTEGRATION.md section 3.
pub mod rule;
pub mod sys;

use std::any::Any;
use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::sync::Arc;

use arrow::array::{
    Array, ArrayRef, AsArray, Float64Array, Int32Array, Int64Array, ListBuilder, RecordBatch, StringArray, StringBuilder,
    StructArray, UInt64Array,
};
use arrow::datatypes::{DataType, Field, Schema, SchemaRef};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::common::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
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

/// Owns the device context + the immutable fused plan; Send + Sync (the C side locks internally).
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

/// Which of the fused shapes (BASELINE.json configs 2-5 + the range form) a `GpuFilterAggExec` runs, with what the state
/// batch needs to become the replaced node's output.
#[derive(Debug, Clone)]
pub enum Shape {
    /// `chrom = lit AND pos BETWEEN a AND b`, COUNT(*): no group column
    RegionCount,
    /// `bam_region_filter(...)`, COUNT(*)
    OverlapCount,
    /// `flag & M = V AND CAST(mapping_quality AS INT) >= q`, COUNT(*) GROUP BY reference (Utf8, nullable)
    FlagMapqGroupCount,
    /// `info.F <op> lit`, AVG(qual), COUNT(*) GROUP BY filter (List<Utf8>); `info_field` = F
    CmpAvgByGroup { info_field: String },
    /// per-position histogram of quality_scores
    QualPosHist,
}

/// Where the rows of a partition come from.
#[derive(Debug, Clone)]
pub enum Source {
    /// the library opens these local files itself (`exon_hip_scan_open`, gpu_parse = 1); one Vec per partition
    Files { format: i32, groups: Vec<Vec<String>>, region: Option<String>, use_index: bool },
    /// the child scan's batches are pushed (`exon_hip_stream_push`); key columns are interned into ids on the way
    ChildBatches,
}

#[derive(Debug)]
pub struct GpuFilterAggExec {
    input: Arc<dyn ExecutionPlan>, // the scan (kept as the child so EXPLAIN still shows where rows come from)
    desc: sys::exon_hip_plan_desc, // fused predicate + aggregates
    shape: Shape,
    source: Source,
    partial_schema: SchemaRef, // schema of the replaced AggregateExec(Partial): group columns, then state fields
    props: PlanProperties,
    gpu: Arc<GpuPlan>,
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
        let want: Vec<DataType> = match &shape {
            Shape::RegionCount | Shape::OverlapCount => vec![DataType::Int64],
            Shape::FlagMapqGroupCount => vec![DataType::Utf8, DataType::Int64],
            Shape::CmpAvgByGroup { .. } => vec![
                DataType::List(Arc::new(Field::new("item", DataType::Utf8, true))),
                DataType::UInt64,
                DataType::Float64,
                DataType::Int64,
            ],
            Shape::QualPosHist => vec![DataType::Int32, DataType::Int32, DataType::Int64],
        };
        let got: Vec<&DataType> = partial_schema.fields().iter().map(|f| f.data_type()).collect();
        if got.len() != want.len() || got.iter().zip(&want).any(|(g, w)| !g.equals_datatype(w)) {
            return Err(DataFusionError::Plan(format!(
                "GpuFilterAggExec: the partial aggregate's schema {got:?} is not the state layout {want:?} of {shape:?}"
            )));
        }
        if let Source::Files { groups, .. } = &source {
            if groups.len() != input.properties().output_partitioning().partition_count() {
                return Err(DataFusionError::Plan("GpuFilterAggExec: one file group per input partition expected".into()));
            }
        }
        let gpu = Arc::new(GpuPlan::try_new(device, &desc)?);
        // same partitioning as the scan; one state batch per partition, emitted once
        let props = PlanProperties::new(
            datafusion::physical_expr::EquivalenceProperties::new(partial_schema.clone()),
            input.properties().output_partitioning().clone(),
            datafusion::physical_plan::execution_plan::EmissionType::Final,
            datafusion::physical_plan::execution_plan::Boundedness::Bounded,
        );
        Ok(Self { input, desc, shape, source, partial_schema, props, gpu })
    }
}

impl DisplayAs for GpuFilterAggExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        let src = match &self.source {
            Source::Files { .. } => "files->HBM",
            Source::ChildBatches => "child batches",
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
        let next = ids.len() as i32;
        let id = *ids.entry(k.clone()).or_insert_with(|| {
            keys.0.push(k);
            next
        });
        out.push(id);
    }
    let mut cols = batch.columns().to_vec();
    cols[key_col] = Arc::new(Int32Array::from(out));
    let mut fields: Vec<Field> = batch.schema().fields().iter().map(|f| f.as_ref().clone()).collect();
    fields[key_col] = Field::new(fields[key_col].name(), DataType::Int32, false);
    RecordBatch::try_new(Arc::new(Schema::new(fields)), cols).map_err(DataFusionError::from)
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
            gpu: self.gpu.clone(),
        }))
    }

    /// One HIP stream per partition; partitions = file groups (regroup_files_by_size), one GPU each in a multi-GPU job.
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let gpu = self.gpu.clone();
        let shape = self.shape.clone();
        let schema = self.partial_schema.clone();
        let out_schema = schema.clone();
        let source = self.source.clone();
        let key_col = self.desc.columns[2] as usize; // K3 / K4: the group key is the third operator argument
        let mut input = match &source {
            Source::ChildBatches => Some(self.input.execute(partition, ctx)?),
            Source::Files { .. } => None,
        };
        let fut = async move {
            let mut s: *mut sys::exon_hip_stream = std::ptr::null_mut();
            check(gpu.ctx, unsafe { sys::exon_hip_stream_open(gpu.plan, partition as i32, &mut s) })?;
            let mut keys = Keys(Vec::new());
            match &source {
                Source::Files { format, groups, region, use_index } => {
                    let info = match &shape {
                        Shape::CmpAvgByGroup { info_field } => Some(CString::new(info_field.as_str()).unwrap()),
                        _ => None,
                    };
                    let region_c = region.as_ref().map(|r| CString::new(r.as_str()).unwrap());
                    for path in &groups[partition] {
                        let opt = sys::exon_hip_scan_options {
                            format: *format,
                            compression: sys::EXON_HIP_COMPRESSION_AUTO,
                            batch_size: 0,
                            info_field: info.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()),
                            region: region_c.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()),
                            use_index: *use_index as i32,
                            gpu_parse: 1,
                        };
                        let cpath = CString::new(path.as_str()).unwrap();
                        let mut scan = std::ptr::null_mut();
                        check(std::ptr::null(), unsafe { sys::exon_hip_scan_open(cpath.as_ptr(), &opt, &mut scan) })?;
                        let mut rows = 0i64;
                        let rc = unsafe { sys::exon_hip_stream_consume_scan(s, scan, &mut rows) };
                        // dictionary of the key column (VCF: scan column 3 = filter; BAM: 2 = reference), in id order;
                        // every file of the partition interns into the same scan-side order because ids are handed out
                        // in order of first appearance per FILE: one file per partition keeps this exact, several files
                        // need `exon_hip_scan_dictionary_intern` of the first file's names before consuming the next.
                        let dict_col = match shape {
                            Shape::CmpAvgByGroup { .. } => 3,
                            _ => 2,
                        };
                        let mut n = 0i32;
                        if rc >= 0 && unsafe { sys::exon_hip_scan_dictionary_size(scan, dict_col, &mut n) } == 0 {
                            for id in keys.0.len() as i32..n {
                                let mut p = std::ptr::null();
                                unsafe { sys::exon_hip_scan_dictionary_value(scan, dict_col, id, &mut p) };
                                keys.0.push(Some(unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()));
                            }
                        }
                        unsafe { sys::exon_hip_scan_close(scan) };
                        check(gpu.ctx, rc)?;
                    }
                }
                Source::ChildBatches => {
                    let mut ids = HashMap::new();
                    let input = input.as_mut().unwrap();
                    while let Some(batch) = input.next().await {
                        let mut batch: RecordBatch = batch?;
                        if matches!(shape, Shape::FlagMapqGroupCount | Shape::CmpAvgByGroup { .. }) {
                            batch = intern_keys(&batch, key_col, &mut ids, &mut keys)?;
                        }
                        let (mut arr, _sch) = to_ffi(&StructArray::from(batch).to_data())?; // zero-copy export
                        // the batch is MOVED: the library calls arr.release exactly once; `_sch` stays ours and is
                        // released by its Drop at the end of this iteration
                        check(gpu.ctx, unsafe { sys::exon_hip_stream_push(s, &mut arr as *mut FFI_ArrowArray) })?;
                        std::mem::forget(arr);
                    }
                }
            }
            let mut out = FFI_ArrowArray::empty();
            let mut out_s = FFI_ArrowSchema::empty();
            let rc = unsafe { sys::exon_hip_stream_finish_arrow(s, &mut out, &mut out_s) };
            unsafe { sys::exon_hip_stream_close(s) };
            check(gpu.ctx, rc)?;
            let data = unsafe { from_ffi(out, &out_s) }?; // takes ownership of `out`; `out_s` is released by its Drop
            state_to_partial(&shape, &StructArray::from(data), &keys, &schema)
        };
        Ok(Box::pin(RecordBatchStreamAdapter::new(out_schema, futures::stream::once(fut))))
    }
}

// keep the unused-import lint quiet for the state column types named in the docs above
#[allow(dead_code)]
fn _state_types(_: &UInt64Array, _: &Float64Array, _: &Int64Array) {}
