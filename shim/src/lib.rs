//! exon-hip: DataFusion glue for the MI355X filter+aggregate path.  NOT COMPILED HERE (no Rust toolchain
//! in the build image); written against datafusion 44 / arrow 53 as pinned by the reference's Cargo.lock.
//!
//! * `GpuFilterAggExec` is an `ExecutionPlan` that consumes the child scan's RecordBatches, pushes them
//!   through `exon_hip_stream_push` (Arrow C Data Interface, zero-copy export with `arrow::ffi::to_ffi`)
//!   and emits ONE batch per partition: the partial-aggregate state that `AggregateExec(Final)` merges.
//! * `GpuFilterAggRule` is the `PhysicalOptimizerRule` that substitutes it for
//!   `AggregateExec(Partial) <- [CoalesceBatchesExec] <- FilterExec <- {VCFScan|BAMScan|FASTQScan}`
//!   when the predicate/aggregates match one of the four fused shapes.
//! * Registration does not touch exon-core: `ExonSession::new(ctx)` accepts any SessionContext
//!   (exon-core/src/session_context/exon_context_ext.rs:103-112).
pub mod sys;

use std::any::Any;
use std::ffi::CStr;
use std::sync::Arc;

use arrow::array::{Array, RecordBatch, StructArray};
use arrow::datatypes::SchemaRef;
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::common::{DataFusionError, Result};
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, PlanProperties};
use futures::StreamExt;

fn check(ctx: *const sys::exon_hip_ctx, rc: i32) -> Result<()> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(sys::exon_hip_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(DataFusionError::External(format!("exon_hip status {rc}: {msg}").into()))
}

/// Owns the device context + the immutable fused plan; Send + Sync (the C side locks internally).
pub struct GpuPlan { ctx: *mut sys::exon_hip_ctx, plan: *mut sys::exon_hip_plan }
unsafe impl Send for GpuPlan {}
unsafe impl Sync for GpuPlan {}
impl Drop for GpuPlan {
    fn drop(&mut self) { unsafe { sys::exon_hip_plan_destroy(self.plan); sys::exon_hip_ctx_destroy(self.ctx); } }
}

#[derive(Debug)]
pub struct GpuFilterAggExec {
    input: Arc<dyn ExecutionPlan>,     // VCFScan / BAMScan / FASTQScan (device-layout projection)
    desc: sys::exon_hip_plan_desc,     // fused predicate + aggregates
    state_schema: SchemaRef,           // DataFusion partial-state schema of the replaced AggregateExec(Partial)
    props: PlanProperties,
    gpu: Arc<GpuPlan>,
}

impl DisplayAs for GpuFilterAggExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "GpuFilterAggExec: kind={}", self.desc.kind)
    }
}

impl ExecutionPlan for GpuFilterAggExec {
    fn name(&self) -> &str { "GpuFilterAggExec" }
    fn as_any(&self) -> &dyn Any { self }
    fn properties(&self) -> &PlanProperties { &self.props }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> { vec![&self.input] }
    fn with_new_children(self: Arc<Self>, c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(Arc::new(Self { input: c[0].clone(), desc: self.desc, state_schema: self.state_schema.clone(),
                           props: self.props.clone(), gpu: self.gpu.clone() }))
    }

    /// One HIP stream per partition; partitions = file groups (regroup_files_by_size), one GPU each.
    fn execute(&self, partition: usize, ctx: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let mut input = self.input.execute(partition, ctx)?;
        let gpu = self.gpu.clone();
        let schema = self.state_schema.clone();
        let out_schema = schema.clone();
        let fut = async move {
            let mut s: *mut sys::exon_hip_stream = std::ptr::null_mut();
            check(gpu.ctx, unsafe { sys::exon_hip_stream_open(gpu.plan, partition as i32, &mut s) })?;
            while let Some(batch) = input.next().await {
                let batch: RecordBatch = batch?;
                let (mut arr, _sch) = to_ffi(&StructArray::from(batch).to_data())?;   // zero-copy export
                check(gpu.ctx, unsafe { sys::exon_hip_stream_push(s, &mut arr as *mut FFI_ArrowArray) })?;
                std::mem::forget(arr);                                                // moved to the library
            }
            let mut out = FFI_ArrowArray::empty();
            let mut out_s = FFI_ArrowSchema::empty();
            check(gpu.ctx, unsafe { sys::exon_hip_stream_finish_arrow(s, &mut out, &mut out_s) })?;
            unsafe { sys::exon_hip_stream_close(s) };
            let data = unsafe { from_ffi(out, &out_s) }?;
            // columns are renamed positionally to DataFusion's state field names (`state_schema`)
            let st = StructArray::from(data);
            RecordBatch::try_new(schema, st.columns().to_vec()).map_err(DataFusionError::from)
        };
        Ok(Box::pin(RecordBatchStreamAdapter::new(out_schema, futures::stream::once(fut))))
    }
}
