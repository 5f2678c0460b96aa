//! `GpuFilterAggRule`: substitutes [`GpuFilterAggExec`] for
//! `AggregateExec(Partial) <- [CoalesceBatchesExec] <- FilterExec <- {VCFScan | BAMScan | SAMScan | CRAMScan | FASTQScan}`.
//! NOT COMPILED HERE (see lib.rs).  Mirrors the recursion of the reference's own rule
//! (exon-core/src/physical_optimizer/chrom_optimizer_rule.rs:26-64) and the plan shape its tests assert
//! (exon-core/src/datasources/vcf/table_provider.rs:571-611).
//!
//! Shapes:
//!   C2  WHERE chrom = 'c' AND pos >= a AND pos <= b                      SELECT COUNT(*)
//!   C3  WHERE flag & M = V AND CAST(mapping_quality AS INT) >= q          SELECT reference, COUNT(*) GROUP BY reference
//!   C4  WHERE info."F" <op> lit                                           SELECT filter, AVG(qual), COUNT(*) GROUP BY filter
//!   C6  WHERE reference = 'r' AND start <= b AND "end" >= a                SELECT COUNT(*)
//! Anything else is returned unchanged: DataFusion's CPU operators run as before.
use std::sync::Arc;

use datafusion::common::tree_node::Transformed;
use datafusion::common::{Result, ScalarValue};
use datafusion::config::ConfigOptions;
use datafusion::logical_expr::Operator;
use datafusion::physical_expr::expressions::{BinaryExpr, CastExpr, Column, Literal};
use datafusion::physical_expr::PhysicalExpr;
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::coalesce_batches::CoalesceBatchesExec;
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::{with_new_children_if_necessary, ExecutionPlan};
use exon::datasources::bam::BAMScan;
use exon::datasources::cram::CRAMScan;
use exon::datasources::sam::SAMScan;
use exon::datasources::vcf::VCFScan;

use crate::{sys, GpuFilterAggExec, Shape, Source};

#[derive(Debug)]
pub struct GpuFilterAggRule {
    device: i32,
}
impl GpuFilterAggRule {
    pub fn new(device: i32) -> Self {
        Self { device }
    }
}

/// `a AND b AND ...` flattened
fn conjuncts(e: &Arc<dyn PhysicalExpr>, out: &mut Vec<Arc<dyn PhysicalExpr>>) {
    if let Some(b) = e.as_any().downcast_ref::<BinaryExpr>() {
        if *b.op() == Operator::And {
            conjuncts(b.left(), out);
            conjuncts(b.right(), out);
            return;
        }
    }
    out.push(e.clone());
}
fn col_name(e: &Arc<dyn PhysicalExpr>) -> Option<(String, usize)> {
    e.as_any().downcast_ref::<Column>().map(|c| (c.name().to_string(), c.index()))
}
fn lit(e: &Arc<dyn PhysicalExpr>) -> Option<ScalarValue> {
    e.as_any().downcast_ref::<Literal>().map(|l| l.value().clone())
}
fn lit_i64(e: &Arc<dyn PhysicalExpr>) -> Option<i64> {
    match lit(e)? {
        ScalarValue::Int64(Some(v)) => Some(v),
        ScalarValue::Int32(Some(v)) => Some(v as i64),
        ScalarValue::UInt64(Some(v)) => i64::try_from(v).ok(),
        _ => None,
    }
}
/// `col <op> literal` (either order is normalised to column-on-the-left)
fn cmp(e: &Arc<dyn PhysicalExpr>) -> Option<(Arc<dyn PhysicalExpr>, Operator, Arc<dyn PhysicalExpr>)> {
    let b = e.as_any().downcast_ref::<BinaryExpr>()?;
    if lit(b.right()).is_some() {
        return Some((b.left().clone(), *b.op(), b.right().clone()));
    }
    if lit(b.left()).is_some() {
        let flipped = match b.op() {
            Operator::Lt => Operator::Gt,
            Operator::LtEq => Operator::GtEq,
            Operator::Gt => Operator::Lt,
            Operator::GtEq => Operator::LtEq,
            o => *o,
        };
        return Some((b.right().clone(), flipped, b.left().clone()));
    }
    None
}
fn cmp_op(op: Operator) -> Option<i32> {
    Some(match op {
        Operator::Gt => sys::EXON_HIP_GT,
        Operator::GtEq => sys::EXON_HIP_GE,
        Operator::Lt => sys::EXON_HIP_LT,
        Operator::LtEq => sys::EXON_HIP_LE,
        Operator::Eq => sys::EXON_HIP_EQ,
        Operator::NotEq => sys::EXON_HIP_NE,
        _ => return None,
    })
}
/// strips `CAST(x AS Float64 / Int32 / Int64)` (DataFusion's type coercion inserts them around Float32 / Utf8 columns)
fn uncast(e: &Arc<dyn PhysicalExpr>) -> Arc<dyn PhysicalExpr> {
    match e.as_any().downcast_ref::<CastExpr>() {
        Some(c) => uncast(c.expr()),
        None => e.clone(),
    }
}

/// The aggregate list must be exactly COUNT(*) (+ AVG(qual)) in the order the state layout holds them.
fn aggregates_are(agg: &AggregateExec, names: &[&str]) -> bool {
    let got: Vec<String> = agg.aggr_expr().iter().map(|a| a.fun().name().to_lowercase()).collect();
    got.len() == names.len() && got.iter().zip(names).all(|(g, w)| g == w)
}

struct Matched {
    desc: sys::exon_hip_plan_desc,
    shape: Shape,
    source: Source,
    scan: Arc<dyn ExecutionPlan>,
}

/// local files of every partition, or None when a file lives on another object store
fn local_groups(scan: &VCFScan) -> Option<Vec<Vec<String>>> {
    let cfg = scan.base_config();
    if cfg.object_store_url.as_str() != "file:///" {
        return None;
    }
    Some(
        cfg.file_groups
            .iter()
            .map(|g| g.iter().map(|f| format!("/{}", f.object_meta.location)).collect())
            .collect(),
    )
}

fn match_vcf(agg: &AggregateExec, pred: &Arc<dyn PhysicalExpr>, scan_plan: &Arc<dyn ExecutionPlan>, scan: &VCFScan) -> Option<Matched> {
    let mut parts = Vec::new();
    conjuncts(pred, &mut parts);
    let schema = scan_plan.schema();
    let group = agg.group_expr().expr();
    // ---- C4: one comparison on info.<F>, GROUP BY filter, AVG(qual) + COUNT(*)
    if parts.len() == 1 && group.len() == 1 && aggregates_are(agg, &["avg", "count"]) {
        let (l, op, r) = cmp(&parts[0])?;
        let thr = match lit(&r)? {
            ScalarValue::Float64(Some(v)) => v,
            ScalarValue::Float32(Some(v)) => v as f64,
            _ => return None,
        };
        // info."F" is a GetFieldFunc over the `info` struct column after `SET exon.vcf_parse_info = true`
        // (exon-core/src/datasources/vcf/schema_builder.rs:197-249): its display form is `info[F]`
        let shown = uncast(&l).to_string();
        let field = shown.strip_prefix("info@")?.split('[').nth(1)?.trim_end_matches(']').to_string();
        let (gname, _) = col_name(&group[0].0)?;
        if gname != "filter" {
            return None;
        }
        let mut desc = sys::exon_hip_plan_desc::default();
        desc.kind = sys::EXON_HIP_PLAN_CMP_AVG_BY_GROUP;
        desc.n_groups = sys::EXON_HIP_MAX_GROUPS; // distinct FILTER lists of a file: far below the LDS table size
        desc.cmp_op = cmp_op(op)?;
        desc.threshold = thr;
        desc.columns = [4, 2, 3, 0]; // scan column order of exon_hip_scan_*: chrom pos qual filter info.<F>
        let source = match local_groups(scan) {
            Some(groups) => Source::Files { format: sys::EXON_HIP_FORMAT_VCF, groups, region: None, use_index: false },
            None => {
                // child batches: info.F, qual, filter by their indexes in the scan's projected schema
                desc.columns = [schema.index_of("info").ok()? as i32, schema.index_of("qual").ok()? as i32, schema.index_of("filter").ok()? as i32, 0];
                Source::ChildBatches
            }
        };
        return Some(Matched { desc, shape: Shape::CmpAvgByGroup { info_field: field }, source, scan: scan_plan.clone() });
    }
    // ---- C2: chrom = lit AND pos >= a AND pos <= b, COUNT(*)
    if group.is_empty() && aggregates_are(agg, &["count"]) {
        let (mut chrom, mut a, mut b) = (None, 1i64, sys::EXON_HIP_REGION_OPEN_END);
        for p in &parts {
            let (l, op, r) = cmp(p)?;
            let (name, _) = col_name(&uncast(&l))?;
            match (name.as_str(), op) {
                ("chrom", Operator::Eq) => match lit(&r)? {
                    ScalarValue::Utf8(Some(s)) | ScalarValue::LargeUtf8(Some(s)) => chrom = Some(s),
                    _ => return None,
                },
                ("pos", Operator::GtEq) => a = a.max(lit_i64(&r)?),
                ("pos", Operator::Gt) => a = a.max(lit_i64(&r)? + 1),
                ("pos", Operator::LtEq) => b = b.min(lit_i64(&r)?),
                ("pos", Operator::Lt) => b = b.min(lit_i64(&r)? - 1),
                _ => return None,
            }
        }
        let chrom = chrom?;
        let groups = local_groups(scan)?; // the contig's dictionary id comes from the file header: files only
        let mut desc = sys::exon_hip_plan_desc::default();
        desc.kind = sys::EXON_HIP_PLAN_REGION_COUNT;
        desc.region_chrom_id = 0; // resolved per file: the region travels as text and becomes the scan's row mask
        desc.region_start = 1;
        desc.region_end = sys::EXON_HIP_REGION_OPEN_END;
        desc.columns = [0, 1, 0, 0];
        // the interval hit itself is pushed down into the scan (k_region_mask on the GPU decode path); the plan then
        // counts the rows the scan emits.  region_chrom_id = id of `chrom` in the header is set by the Exec per file.
        let region = if b == sys::EXON_HIP_REGION_OPEN_END { format!("{chrom}:{a}") } else { format!("{chrom}:{a}-{b}") };
        let source = Source::Files { format: sys::EXON_HIP_FORMAT_VCF, groups, region: Some(region), use_index: false };
        return Some(Matched { desc, shape: Shape::RegionCount, source, scan: scan_plan.clone() });
    }
    None
}

fn match_bam(agg: &AggregateExec, pred: &Arc<dyn PhysicalExpr>, scan_plan: &Arc<dyn ExecutionPlan>) -> Option<Matched> {
    let mut parts = Vec::new();
    conjuncts(pred, &mut parts);
    let schema = scan_plan.schema();
    let group = agg.group_expr().expr();
    if group.len() != 1 || !aggregates_are(agg, &["count"]) || col_name(&group[0].0)?.0 != "reference" {
        return None;
    }
    let (mut mask, mut value, mut qmin) = (0i32, 0i32, 0i32);
    for p in &parts {
        let (l, op, r) = cmp(p)?;
        let l = uncast(&l);
        if let Some(band) = l.as_any().downcast_ref::<BinaryExpr>() {
            // flag & M = V
            if *band.op() != Operator::BitwiseAnd || op != Operator::Eq || col_name(&uncast(band.left()))?.0 != "flag" {
                return None;
            }
            mask = lit_i64(band.right())? as i32;
            value = lit_i64(&r)? as i32;
        } else if col_name(&l)?.0 == "mapping_quality" && op == Operator::GtEq {
            qmin = lit_i64(&r)? as i32; // CAST(mapping_quality AS INT) >= q
        } else {
            return None;
        }
    }
    let mut desc = sys::exon_hip_plan_desc::default();
    desc.kind = sys::EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT;
    desc.n_groups = sys::EXON_HIP_MAX_GROUPS - 1;
    desc.flag_mask = mask;
    desc.flag_value = value;
    desc.mapq_min = qmin;
    // BAMScan keeps its FileScanConfig private (exon-core/src/datasources/bam/scanner.rs): child batches.  The stream
    // path needs mapping_quality as UInt8 + validity: the Exec casts the Utf8 column before the push.
    desc.columns = [schema.index_of("flag").ok()? as i32, schema.index_of("mapping_quality").ok()? as i32, schema.index_of("reference").ok()? as i32, 0];
    Some(Matched { desc, shape: Shape::FlagMapqGroupCount, source: Source::ChildBatches, scan: scan_plan.clone() })
}

impl GpuFilterAggRule {
    fn rewrite(&self, plan: Arc<dyn ExecutionPlan>) -> Result<Transformed<Arc<dyn ExecutionPlan>>> {
        // children first (same recursion as ExonChromOptimizer)
        let plan = if plan.children().is_empty() {
            plan
        } else {
            let children = plan
                .children()
                .iter()
                .map(|c| self.rewrite((*c).clone()).map(|t| t.data))
                .collect::<Result<Vec<_>>>()?;
            with_new_children_if_necessary(plan, children)?
        };
        let Some(agg) = plan.as_any().downcast_ref::<AggregateExec>() else { return Ok(Transformed::no(plan)) };
        if *agg.mode() != AggregateMode::Partial {
            return Ok(Transformed::no(plan));
        }
        // [CoalesceBatchesExec] <- FilterExec <- scan
        let mut below = agg.input().clone();
        if let Some(c) = below.as_any().downcast_ref::<CoalesceBatchesExec>() {
            below = c.input().clone();
        }
        let Some(filter) = below.as_any().downcast_ref::<FilterExec>() else { return Ok(Transformed::no(plan)) };
        if filter.projection().is_some() {
            return Ok(Transformed::no(plan));
        }
        let scan_plan = filter.input().clone();
        let matched = if let Some(scan) = scan_plan.as_any().downcast_ref::<VCFScan>() {
            match_vcf(agg, filter.predicate(), &scan_plan, scan)
        } else if scan_plan.as_any().downcast_ref::<BAMScan>().is_some()
            || scan_plan.as_any().downcast_ref::<SAMScan>().is_some()
            || scan_plan.as_any().downcast_ref::<CRAMScan>().is_some()
        {
            // SAM, BAM and CRAM share one schema (exon-sam/src/schema_builder.rs:371-402): the same two shapes apply; the
            // child scan's batches are pushed (their FileScanConfig is private)
            match_bam(agg, filter.predicate(), &scan_plan)
        } else {
            None
        };
        let Some(m) = matched else { return Ok(Transformed::no(plan)) };
        match GpuFilterAggExec::try_new(m.scan, m.desc, m.shape, m.source, agg.schema(), self.device) {
            Ok(exec) => Ok(Transformed::yes(Arc::new(exec) as Arc<dyn ExecutionPlan>)),
            // no GPU on this machine, or a schema this build does not cover: keep DataFusion's plan
            Err(_) => Ok(Transformed::no(plan)),
        }
    }
}

impl PhysicalOptimizerRule for GpuFilterAggRule {
    fn optimize(&self, plan: Arc<dyn ExecutionPlan>, _config: &ConfigOptions) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(self.rewrite(plan)?.data)
    }
    fn name(&self) -> &str {
        "exon_hip_gpu_filter_agg"
    }
    fn schema_check(&self) -> bool {
        true // the substituted node reports the replaced AggregateExec(Partial)'s schema
    }
}

#[cfg(test)]
mod tests {
    //! To run on a machine with cargo + an MI355X (mirrors exon-core/src/datasources/vcf/table_provider.rs:571-611):
    use super::*;
    use datafusion::execution::session_state::SessionStateBuilder;
    use datafusion::prelude::SessionContext;
    use exon::ExonSession;

    #[tokio::test]
    async fn plan_shape_c4() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let exon = ExonSession::new_exon()?;
        let state = SessionStateBuilder::new_from_existing(exon.session.state())
            .with_physical_optimizer_rule(Arc::new(GpuFilterAggRule::new(0)))
            .build();
        let ctx = ExonSession::new(SessionContext::new_with_state(state));
        ctx.session.sql("SET exon.vcf_parse_info = true").await?;
        ctx.session.sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf'").await?;
        let df = ctx.session.sql("SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info.\"MQ0F\" > -1 GROUP BY filter").await?;
        let plan = ctx.session.state().create_physical_plan(df.logical_plan()).await?;
        let shown = datafusion::physical_plan::displayable(plan.as_ref()).indent(true).to_string();
        assert!(shown.contains("GpuFilterAggExec"), "{shown}");
        assert!(shown.contains("AggregateExec: mode=Final"), "{shown}");
        let rows: usize = df.collect().await?.iter().map(|b| b.num_rows()).sum();
        assert_eq!(rows, 1); // all 621 records carry the empty FILTER list
        Ok(())
    }
}
