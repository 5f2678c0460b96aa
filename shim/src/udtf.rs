//! `fastq_quality_histogram('<path>' [, 'gzip'])`: BASELINE.json config 5 behind the reference's table-function surface.
//! NOT COMPILED HERE (see lib.rs).
//!
//! The per-position quality histogram has no single DataFusion operator: in SQL it is
//! `quality_scores_to_list` (exon-core/src/udfs/sequence/quality_score_string_to_list.rs:56-117) + `unnest` + `GROUP BY
//! (position, score)`, a five-node plan whose exact shape depends on the DataFusion version and which nothing in the
//! reference's tests pins.  Instead of pattern-matching it, the shape is exposed the way the reference exposes its own
//! scans -- a `TableFunctionImpl` registered with `SessionContext::register_udtf`
//! (exon-core/src/datasources/fastq/udtf.rs:46-70, exon-core/src/session_context/exon_context_ext.rs:216-223) -- whose
//! `TableProvider::scan` returns a [`GpuFilterAggExec`] over the local files: FASTQ text goes to HBM as it is, record
//! splitting (`exon_hip_fastq_parser_*`) and the histogram kernel run on the GPU.
//!
//! Output schema: `position Int32` (0-based), `quality_score Int32` (= byte - 33, the UDF's rule :83-86), `count Int64`;
//! one row per observed (position, score) and FILE -- several files give several partial rows per key, so aggregate with
//! `SELECT position, quality_score, SUM(count) ... GROUP BY 1, 2` when a directory is scanned.
use std::any::Any;
use std::sync::Arc;

use arrow::datatypes::{DataType, Field, Schema, SchemaRef};
use async_trait::async_trait;
use datafusion::catalog::{Session, TableFunctionImpl};
use datafusion::common::{DataFusionError, Result, ScalarValue};
use datafusion::datasource::{TableProvider, TableType};
use datafusion::logical_expr::Expr;
use datafusion::physical_plan::empty::EmptyExec;
use datafusion::physical_plan::ExecutionPlan;
use datafusion::prelude::SessionContext;

use crate::{sys, GpuFilterAggExec, Shape, Source};

/// longest read the histogram keeps (positions >= LMAX are an error of the library: status bit 8)
const LMAX: i32 = 1024;

fn histogram_schema() -> SchemaRef {
    Arc::new(Schema::new(vec![
        Field::new("position", DataType::Int32, false),
        Field::new("quality_score", DataType::Int32, false),
        Field::new("count", DataType::Int64, false),
    ]))
}

#[derive(Debug)]
pub struct FastqQualityHistogram {
    device: i32,
}
impl FastqQualityHistogram {
    pub fn new(device: i32) -> Self {
        Self { device }
    }
}

/// `ctx.register_udtf("fastq_quality_histogram", ...)`, next to the reference's own `fastq_scan`
pub fn register(ctx: &SessionContext, device: i32) {
    ctx.register_udtf("fastq_quality_histogram", Arc::new(FastqQualityHistogram::new(device)));
}

impl TableFunctionImpl for FastqQualityHistogram {
    fn call(&self, args: &[Expr]) -> Result<Arc<dyn TableProvider>> {
        let path = match args.first() {
            Some(Expr::Literal(ScalarValue::Utf8(Some(p)))) => p.clone(),
            _ => return Err(DataFusionError::Plan("fastq_quality_histogram('<path>'): the first argument must be a string literal".into())),
        };
        // a local file or a directory of *.fastq / *.fq (optionally .gz / .bgz: the library sniffs the compression)
        let meta = std::fs::metadata(&path).map_err(|e| DataFusionError::Plan(format!("{path}: {e}")))?;
        let mut files = Vec::new();
        if meta.is_dir() {
            for entry in std::fs::read_dir(&path).map_err(|e| DataFusionError::Plan(format!("{path}: {e}")))? {
                let p = entry.map_err(|e| DataFusionError::Plan(e.to_string()))?.path();
                let name = p.file_name().and_then(|n| n.to_str()).unwrap_or("");
                if [".fastq", ".fq", ".fastq.gz", ".fq.gz", ".fastq.bgz", ".fq.bgz"].iter().any(|ext| name.ends_with(ext)) {
                    files.push(p.to_string_lossy().into_owned());
                }
            }
            files.sort();
        } else {
            files.push(path);
        }
        Ok(Arc::new(HistogramTable { files, device: self.device }))
    }
}

#[derive(Debug)]
struct HistogramTable {
    files: Vec<String>,
    device: i32,
}

#[async_trait]
impl TableProvider for HistogramTable {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn schema(&self) -> SchemaRef {
        histogram_schema()
    }
    fn table_type(&self) -> TableType {
        TableType::Base
    }
    async fn scan(
        &self,
        _state: &dyn Session,
        projection: Option<&Vec<usize>>,
        _filters: &[Expr],
        _limit: Option<usize>,
    ) -> Result<Arc<dyn ExecutionPlan>> {
        let schema = histogram_schema();
        let mut desc = sys::exon_hip_plan_desc::default();
        desc.kind = sys::EXON_HIP_PLAN_QUAL_POS_HIST;
        desc.lmax = LMAX;
        desc.columns = [3, 0, 0, 0]; // scan column order of the FASTQ decoders: name description sequence quality_scores
        // one partition holding every file: the placeholder child only carries the partitioning (1) into the node
        let child: Arc<dyn ExecutionPlan> = Arc::new(EmptyExec::new(schema.clone()));
        let source = Source::Files { format: sys::EXON_HIP_FORMAT_FASTQ, groups: vec![self.files.clone()], region: None, use_index: false };
        let exec: Arc<dyn ExecutionPlan> =
            Arc::new(GpuFilterAggExec::try_new(child, desc, Shape::QualPosHist, source, schema, self.device)?);
        match projection {
            None => Ok(exec),
            Some(p) => {
                use datafusion::physical_expr::expressions::Column;
                use datafusion::physical_plan::projection::ProjectionExec;
                let s = exec.schema();
                let exprs = p
                    .iter()
                    .map(|&i| (Arc::new(Column::new(s.field(i).name(), i)) as Arc<dyn datafusion::physical_expr::PhysicalExpr>, s.field(i).name().clone()))
                    .collect();
                Ok(Arc::new(ProjectionExec::try_new(exprs, exec)?))
            }
        }
    }
}
