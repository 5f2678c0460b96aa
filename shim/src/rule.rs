//! `GpuFilterAggRule`: substitutes [`GpuFilterAggExec`] for the partial aggregates the reference really plans.
//! NOT COMPILED HERE (see lib.rs).  Mirrors the recursion of the reference's own rule
//! (exon-core/src/physical_optimizer/chrom_optimizer_rule.rs:26-64) and the plan shapes its tests assert
//! (exon-core/src/datasources/vcf/table_provider.rs:571-611).
//!
//! Plan shapes (`A` = `AggregateExec(mode=Partial)`, `F` = `FilterExec` -- with or without an embedded `projection=[..]`;
//! `[C]` = any run of the nodes DataFusion's own rules put in between and that change neither rows nor values:
//! `CoalesceBatchesExec`, `RepartitionExec(RoundRobinBatch)` and a `ProjectionExec` of plain columns.  `new_exon_config` sets
//! `target_partitions = num_cpus` and leaves round-robin repartitioning on (exon-core/src/config/mod.rs:27-45), so with
//! fewer files than cores `EnforceDistribution` puts `RepartitionExec(RoundRobinBatch(n))` between `F` and a one-partition
//! scan, and between `A` and an indexed scan; this rule runs AFTER the default rules and must see
//! through those nodes -- `[C]` is peeled both above and below `F`):
//!
//! | id  | SQL                                                                  | physical plan below `A`                         | matcher |
//! |-----|----------------------------------------------------------------------|--------------------------------------------------|---------|
//! | C2  | `COUNT(*) WHERE chrom = 'c' AND pos >= a AND pos <= b`                | `[C] <- F <- VCFScan`                            | `match_vcf_filter` |
//! | C2r | `COUNT(*) WHERE vcf_region_filter('c:a-b', chrom, pos)` (pushed down `Exact`) | `IndexedVCFScanner` (NO FilterExec)      | `match_indexed_vcf` |
//! | C3  | `reference, COUNT(*) WHERE flag & M = V AND CAST(mapping_quality AS INT) >= q GROUP BY reference` | `[C] <- F <- BAMScan / SAMScan / CRAMScan / IndexedBAMScan` | `match_alignment_filter` |
//! | C4  | `filter, AVG(qual), COUNT(*) WHERE info."F" <op> lit GROUP BY filter` | `[C] <- F <- VCFScan / IndexedVCFScanner`        | `match_vcf_filter` |
//! | C6  | `COUNT(*) WHERE reference = 'r' AND start <= b AND "end" >= a`         | `[C] <- F <- BAMScan / SAMScan / CRAMScan`       | `match_alignment_filter` |
//! | C6r | `COUNT(*) WHERE bam_region_filter('r:a-b', reference, start, end)`     | `IndexedBAMScan` (NO FilterExec)                 | `match_indexed_bam` |
//! | C5  | per-position quality histogram                                        | no operator to match: `fastq_quality_histogram()` | `udtf.rs` |
//!
//! Anything else is returned unchanged: DataFusion's CPU operators run as before.
//!
//! Scans are recognised by `ExecutionPlan::name()` where their Rust type is not needed (`CRAMScan` lives in a
//! `pub(crate)` module, exon-core/src/datasources/cram/mod.rs:15-20, and cannot be imported) and by downcast where a public
//! accessor is used (`VCFScan::base_config`, vcf/scanner.rs:75; `IndexedVCFScanner::base_config`, vcf/indexed_scanner.rs:71).
use std::sync::Arc;

use datafusion::common::tree_node::Transformed;
use datafusion::common::{Result, ScalarValue};
use datafusion::config::ConfigOptions;
use datafusion::datasource::physical_plan::FileScanConfig;
use datafusion::logical_expr::Operator;
use datafusion::physical_expr::expressions::{BinaryExpr, CastExpr, Column, Literal};
use datafusion::physical_expr::{PhysicalExpr, ScalarFunctionExpr};
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::coalesce_batches::CoalesceBatchesExec;
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::repartition::RepartitionExec;
use datafusion::physical_plan::{with_new_children_if_necessary, ExecutionPlan, Partitioning};
use exon::datasources::bam::IndexedBAMScan;
use exon::datasources::vcf::{IndexedVCFScanner, VCFScan};

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

/// scans with the SAM/BAM schema (exon-sam/src/schema_builder.rs:371-402), by `ExecutionPlan::name()`
/// (bam/scanner.rs:96, sam/scanner.rs:82, cram/scanner.rs:84, bam/indexed_scanner.rs:82, cram/indexed_scanner.rs:82)
const ALIGNMENT_SCANS: [&str; 5] = ["BAMScan", "SAMScan", "CRAMScan", "IndexedBAMScan", "IndexedCRAMScan"];

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
fn lit_str(e: &Arc<dyn PhysicalExpr>) -> Option<String> {
    match lit(e)? {
        ScalarValue::Utf8(Some(s)) | ScalarValue::LargeUtf8(Some(s)) | ScalarValue::Utf8View(Some(s)) => Some(s),
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
/// `info."F"` = the scalar function `get_field(info@N, 'F')` over the `info` struct column (after
/// `SET exon.vcf_parse_info = true`; exon-core/src/datasources/vcf/schema_builder.rs:197-249).  Matched structurally:
/// the node must be a `ScalarFunctionExpr` named `get_field` whose arguments are the Column `info` and a Utf8 literal.
/// (Its `Display` text is not parsed: the physical and logical renderings differ.)
fn info_field_of(e: &Arc<dyn PhysicalExpr>) -> Option<String> {
    let e = uncast(e);
    let f = e.as_any().downcast_ref::<ScalarFunctionExpr>()?;
    if f.name() != "get_field" || f.args().len() != 2 {
        return None;
    }
    let (base, _) = col_name(&f.args()[0])?;
    if base != "info" {
        return None;
    }
    lit_str(&f.args()[1])
}

/// The aggregate list must be exactly these functions, in the order the state layout holds them.
fn aggregates_are(agg: &AggregateExec, names: &[&str]) -> bool {
    let got: Vec<String> = agg.aggr_expr().iter().map(|a| a.fun().name().to_lowercase()).collect();
    got.len() == names.len() && got.iter().zip(names).all(|(g, w)| g == w)
}
/// `AVG(qual)`: the first aggregate's only argument is the Column `qual` (possibly under the Float64 cast)
fn avg_is_over_qual(agg: &AggregateExec) -> bool {
    agg.aggr_expr()
        .first()
        .map(|a| a.expressions())
        .filter(|args| args.len() == 1)
        .and_then(|args| col_name(&uncast(&args[0])))
        .map_or(false, |(n, _)| n == "qual")
}

struct Matched {
    desc: sys::exon_hip_plan_desc,
    shape: Shape,
    source: Source,
    scan: Arc<dyn ExecutionPlan>,
}

/// local files of every partition, or None when a file lives on another object store (the child's batches are pushed then)
fn local_groups(cfg: &FileScanConfig) -> Option<Vec<Vec<String>>> {
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

/// The region of an indexed scanner.  Both keep it in a private field with no accessor
/// (vcf/indexed_scanner.rs:46-47, bam/indexed_scanner.rs:44-45) but derive `Debug` (:34 / :32), so the default build reads
/// it from the derived rendering `region: Region { name: "1", interval: Interval { start: Some(Position(5)), end: None } }`
/// (noodles-core 0.15 `Region` / `Interval` / `Position`, all `#[derive(Debug)]`).  Anything unexpected yields None and the
/// plan is left to DataFusion -- a failed parse can cost the substitution, never a wrong answer.  With the cargo feature
/// `exon-region-accessor` (a two-line `pub fn region(&self) -> &Region` patch to exon-core) the
/// accessor is used instead.
fn region_from_debug(dbg: &str) -> Option<String> {
    let at = dbg.find("region: Region {")?;
    let s = &dbg[at..];
    let name_at = s.find("name: \"")? + 7;
    let name_len = s[name_at..].find('"')?;
    let name = &s[name_at..name_at + name_len];
    if name.is_empty() || name.contains('\\') {
        return None; // an escaped byte in the name: do not guess
    }
    let bound = |key: &str| -> Option<Option<u64>> {
        let k = s.find(key)? + key.len();
        let rest = s[k..].trim_start();
        if rest.starts_with("None") {
            return Some(None);
        }
        let digits = rest.strip_prefix("Some(Position(")?;
        let end = digits.find(')')?;
        digits[..end].parse::<u64>().ok().map(Some)
    };
    let start = bound("start:")?;
    let end = bound("end:")?;
    Some(match (start, end) {
        (None, None) => name.to_string(),
        (Some(a), None) => format!("{name}:{a}"),
        (None, Some(b)) => format!("{name}:1-{b}"),
        (Some(a), Some(b)) => format!("{name}:{a}-{b}"),
    })
}
#[cfg(not(feature = "exon-region-accessor"))]
fn indexed_vcf_region(scan: &IndexedVCFScanner) -> Option<String> {
    region_from_debug(&format!("{scan:?}"))
}
#[cfg(feature = "exon-region-accessor")]
fn indexed_vcf_region(scan: &IndexedVCFScanner) -> Option<String> {
    Some(scan.region().to_string())
}
#[cfg(not(feature = "exon-region-accessor"))]
fn indexed_bam_region(scan: &IndexedBAMScan) -> Option<String> {
    region_from_debug(&format!("{scan:?}"))
}
#[cfg(feature = "exon-region-accessor")]
fn indexed_bam_region(scan: &IndexedBAMScan) -> Option<String> {
    Some(scan.region().to_string())
}
/// local files of an `IndexedBAMScan`: its `base_config` has no accessor either (bam/indexed_scanner.rs:39), so without the
/// feature the scan's batches are pushed instead
#[cfg(not(feature = "exon-region-accessor"))]
fn indexed_bam_groups(_scan: &IndexedBAMScan) -> Option<Vec<Vec<String>>> {
    None
}
#[cfg(feature = "exon-region-accessor")]
fn indexed_bam_groups(scan: &IndexedBAMScan) -> Option<Vec<Vec<String>>> {
    local_groups(scan.base_config())
}

/// C4 over a VCF scan: one comparison on info.<F>, GROUP BY filter, AVG(qual) + COUNT(*).
/// `region`: Some for an `IndexedVCFScanner` (the scan's own pushed-down region rides along into `exon_hip_scan_open`).
fn match_c4(
    agg: &AggregateExec,
    parts: &[Arc<dyn PhysicalExpr>],
    scan_plan: &Arc<dyn ExecutionPlan>,
    cfg: &FileScanConfig,
    region: Option<String>,
) -> Option<Matched> {
    let group = agg.group_expr().expr();
    if parts.len() != 1 || group.len() != 1 || !aggregates_are(agg, &["avg", "count"]) || !avg_is_over_qual(agg) {
        return None;
    }
    let (l, op, r) = cmp(&parts[0])?;
    // Float32 field vs Float64 literal (DataFusion widens the column); Int32 field (INFO Type=Integer,
    // exon-core/src/datasources/vcf/schema_builder.rs:197-205) vs an integer literal: the library compares integers exactly
    // (exon_hip_plan_desc.x_type comes from the file's header in exon_hip_stream_consume_scan); an i64 beyond 2^53 loses
    // bits in `as f64`, but it is beyond int32 too, so the comparison saturates the same way
    let thr = match lit(&r)? {
        ScalarValue::Float64(Some(v)) => v,
        ScalarValue::Float32(Some(v)) => v as f64,
        ScalarValue::Int64(Some(v)) => v as f64,
        ScalarValue::Int32(Some(v)) => v as f64,
        _ => return None,
    };
    let field = info_field_of(&l)?;
    if col_name(&group[0].0)?.0 != "filter" {
        return None;
    }
    let schema = scan_plan.schema();
    let mut desc = sys::exon_hip_plan_desc::default();
    desc.kind = sys::EXON_HIP_PLAN_CMP_AVG_BY_GROUP;
    desc.n_groups = sys::EXON_HIP_MAX_GROUPS; // distinct FILTER lists of a file: far below the LDS table size
    desc.cmp_op = cmp_op(op)?;
    desc.threshold = thr;
    desc.columns = [4, 2, 3, 0]; // scan column order of exon_hip_scan_*: chrom pos qual filter info.<F>
    let use_index = region.is_some();
    let source = match local_groups(cfg) {
        Some(groups) => Source::Files { format: sys::EXON_HIP_FORMAT_VCF, groups, region, use_index },
        None => {
            // child batches (the scanner has applied its region already): info, qual, filter by their indexes in the
            // scan's projected schema.  The `info` child is a Struct: the pushed column must be the F child itself,
            // which only the library's own decoders produce -- without local files the shape is left to DataFusion.
            let _ = schema;
            return None;
        }
    };
    Some(Matched { desc, shape: Shape::CmpAvgByGroup { info_field: field }, source, scan: scan_plan.clone() })
}

/// `chrom = lit AND pos >= a AND pos <= b` -> ("chrom", a, b); any other conjunct: None
fn point_region(parts: &[Arc<dyn PhysicalExpr>]) -> Option<(String, i64, i64)> {
    let (mut chrom, mut a, mut b) = (None, 1i64, sys::EXON_HIP_REGION_OPEN_END);
    for p in parts {
        let (l, op, r) = cmp(p)?;
        let (name, _) = col_name(&uncast(&l))?;
        match (name.as_str(), op) {
            ("chrom", Operator::Eq) => chrom = Some(lit_str(&r)?),
            ("pos", Operator::GtEq) => a = a.max(lit_i64(&r)?),
            ("pos", Operator::Gt) => a = a.max(lit_i64(&r)?.checked_add(1)?),
            ("pos", Operator::LtEq) => b = b.min(lit_i64(&r)?),
            ("pos", Operator::Lt) => b = b.min(lit_i64(&r)?.checked_sub(1)?),
            ("pos", Operator::Eq) => {
                a = a.max(lit_i64(&r)?);
                b = b.min(lit_i64(&r)?);
            }
            _ => return None,
        }
    }
    if a > b || b < 1 {
        return None; // an empty interval (`pos <= 0`, `pos >= 10 AND pos <= 5`): DataFusion answers 0 by itself
    }
    Some((chrom?, a.max(1), b))
}
fn region_text(name: &str, a: i64, b: i64) -> String {
    if b == sys::EXON_HIP_REGION_OPEN_END {
        format!("{name}:{a}")
    } else {
        format!("{name}:{a}-{b}")
    }
}

/// shapes over `F <- VCFScan` / `F <- IndexedVCFScanner`
fn match_vcf_filter(
    agg: &AggregateExec,
    pred: &Arc<dyn PhysicalExpr>,
    scan_plan: &Arc<dyn ExecutionPlan>,
    cfg: &FileScanConfig,
    scan_region: Option<String>,
) -> Option<Matched> {
    let mut parts = Vec::new();
    conjuncts(pred, &mut parts);
    if let Some(m) = match_c4(agg, &parts, scan_plan, cfg, scan_region.clone()) {
        return Some(m);
    }
    // ---- C2: chrom = lit AND pos >= a AND pos <= b, COUNT(*) (plain VCFScan only: an indexed scan has its own region)
    if scan_region.is_some() || !agg.group_expr().expr().is_empty() || !aggregates_are(agg, &["count"]) {
        return None;
    }
    let (chrom, a, b) = point_region(&parts)?;
    let mut desc = sys::exon_hip_plan_desc::default();
    desc.kind = sys::EXON_HIP_PLAN_REGION_COUNT;
    desc.columns = [0, 1, 0, 0];
    match local_groups(cfg) {
        Some(groups) => {
            // the interval hit is pushed down into the scan (k_region_mask on the GPU decode path); the plan then counts the
            // rows the scan emits: region_chrom_id / start / end are set per file by the Exec (lib.rs, `per_file_plan`)
            let source = Source::Files { format: sys::EXON_HIP_FORMAT_VCF, groups, region: Some(region_text(&chrom, a, b)), use_index: false };
            Some(Matched { desc, shape: Shape::RegionCount, source, scan: scan_plan.clone() })
        }
        None => {
            // another object store: the scan's batches are pushed; `chrom` is interned with the literal seeded as id 0
            let schema = scan_plan.schema();
            desc.region_chrom_id = 0;
            desc.region_start = a;
            desc.region_end = b;
            desc.columns = [schema.index_of("chrom").ok()? as i32, schema.index_of("pos").ok()? as i32, 0, 0];
            Some(Matched { desc, shape: Shape::RegionCount, source: Source::ChildBatches { seed_key: Some(chrom) }, scan: scan_plan.clone() })
        }
    }
}

/// C2r: `AggregateExec(Partial) <- IndexedVCFScanner`, COUNT(*) with no group: the region filter was pushed down `Exact`
/// (vcf/table_provider.rs:299-320), so there is no FilterExec and the region lives in the scanner.
fn match_indexed_vcf(agg: &AggregateExec, scan_plan: &Arc<dyn ExecutionPlan>, scan: &IndexedVCFScanner) -> Option<Matched> {
    if !agg.group_expr().expr().is_empty() || !aggregates_are(agg, &["count"]) {
        return None;
    }
    let region = indexed_vcf_region(scan)?;
    let groups = local_groups(scan.base_config())?;
    let mut desc = sys::exon_hip_plan_desc::default();
    desc.kind = sys::EXON_HIP_PLAN_REGION_COUNT;
    desc.columns = [0, 1, 0, 0];
    // use_index: the library plans the tabix chunks itself (host/bgzf_index.h reproduces get_byte_range_for_file,
    // exon-core/src/datasources/indexed_file/indexed_bgzf_file.rs:52-155); only those BGZF blocks cross PCIe
    let source = Source::Files { format: sys::EXON_HIP_FORMAT_VCF, groups, region: Some(region), use_index: true };
    Some(Matched { desc, shape: Shape::RegionCount, source, scan: scan_plan.clone() })
}

/// C6r: `AggregateExec(Partial) <- IndexedBAMScan`, COUNT(*): `bam_region_filter` pushed down (bam/indexed_scanner.rs:34-160).
/// Needs the scanner's files AND region; `base_config` is private there, so this shape is substituted only with the
/// `exon-region-accessor` feature (without it `indexed_bam_groups` is None and the plan stays as it is).
fn match_indexed_bam(agg: &AggregateExec, scan_plan: &Arc<dyn ExecutionPlan>, scan: &IndexedBAMScan) -> Option<Matched> {
    if !agg.group_expr().expr().is_empty() || !aggregates_are(agg, &["count"]) {
        return None;
    }
    let region = indexed_bam_region(scan)?;
    let groups = indexed_bam_groups(scan)?;
    let mut desc = sys::exon_hip_plan_desc::default();
    desc.kind = sys::EXON_HIP_PLAN_OVERLAP_COUNT;
    desc.region_start = 1;
    desc.region_end = sys::EXON_HIP_REGION_OPEN_END;
    desc.columns = [2, 3, 4, 0]; // scan column order of the BAM decoders: name flag reference start end ...
    let source = Source::Files { format: sys::EXON_HIP_FORMAT_BAM, groups, region: Some(region), use_index: true };
    Some(Matched { desc, shape: Shape::OverlapCount, source, scan: scan_plan.clone() })
}

/// shapes over `F <- {BAMScan | SAMScan | CRAMScan | IndexedBAMScan | IndexedCRAMScan}`: the child's batches are pushed
/// (their FileScanConfig is private: bam/scanner.rs:35-50)
fn match_alignment_filter(agg: &AggregateExec, pred: &Arc<dyn PhysicalExpr>, scan_plan: &Arc<dyn ExecutionPlan>) -> Option<Matched> {
    let mut parts = Vec::new();
    conjuncts(pred, &mut parts);
    let schema = scan_plan.schema();
    let group = agg.group_expr().expr();
    // ---- C6: reference = 'r' AND start <= b AND "end" >= a, COUNT(*)  (SemiLazyRecord::intersects as a conjunction,
    //      exon-bam/src/indexed_async_batch_stream.rs:66-87)
    if group.is_empty() && aggregates_are(agg, &["count"]) {
        let (mut reference, mut a, mut b) = (None, 1i64, sys::EXON_HIP_REGION_OPEN_END);
        for p in &parts {
            let (l, op, r) = cmp(p)?;
            let (name, _) = col_name(&uncast(&l))?;
            match (name.as_str(), op) {
                ("reference", Operator::Eq) => reference = Some(lit_str(&r)?),
                ("start", Operator::LtEq) => b = b.min(lit_i64(&r)?),
                ("start", Operator::Lt) => b = b.min(lit_i64(&r)?.checked_sub(1)?),
                ("end", Operator::GtEq) => a = a.max(lit_i64(&r)?),
                ("end", Operator::Gt) => a = a.max(lit_i64(&r)?.checked_add(1)?),
                _ => return None,
            }
        }
        let reference = reference?;
        if a > b {
            return None; // an empty interval: exon_hip_plan_create rejects it; DataFusion answers 0 by itself
        }
        let mut desc = sys::exon_hip_plan_desc::default();
        desc.kind = sys::EXON_HIP_PLAN_OVERLAP_COUNT;
        desc.region_chrom_id = 0; // the literal is seeded as id 0 of the interner
        desc.region_start = a;
        desc.region_end = b;
        desc.columns = [schema.index_of("reference").ok()? as i32, schema.index_of("start").ok()? as i32, schema.index_of("end").ok()? as i32, 0];
        return Some(Matched { desc, shape: Shape::OverlapCount, source: Source::ChildBatches { seed_key: Some(reference) }, scan: scan_plan.clone() });
    }
    // ---- C3: flag & M = V AND CAST(mapping_quality AS INT) >= q, COUNT(*) GROUP BY reference
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
            mask = i32::try_from(lit_i64(band.right())?).ok()?;
            value = i32::try_from(lit_i64(&r)?).ok()?;
        } else if col_name(&l)?.0 == "mapping_quality" && matches!(op, Operator::GtEq | Operator::Gt) {
            // CAST(mapping_quality AS INT) >= q   (> q is >= q + 1)
            let q = lit_i64(&r)?.checked_add((op == Operator::Gt) as i64)?;
            qmin = i32::try_from(q.clamp(0, 256)).ok()?;
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
    // `mapping_quality` is Utf8 in the scan's schema (exon-sam/src/schema_builder.rs:392): GpuFilterAggExec::execute
    // converts it to UInt8 + validity (`mapq_to_u8` in lib.rs) before every push
    desc.columns = [schema.index_of("flag").ok()? as i32, schema.index_of("mapping_quality").ok()? as i32, schema.index_of("reference").ok()? as i32, 0];
    Some(Matched { desc, shape: Shape::FlagMapqGroupCount, source: Source::ChildBatches { seed_key: None }, scan: scan_plan.clone() })
}

/// Skips the nodes DataFusion's default rules insert between the partial aggregate, the filter and the scan and that
/// change neither the set of rows nor their values: `CoalesceBatchesExec` (the `CoalesceBatches` rule wraps every
/// FilterExec), `RepartitionExec` with `Partitioning::RoundRobinBatch` (`EnforceDistribution` adds one on top of a scan with
/// fewer partitions than `target_partitions`; a Hash repartition is NOT skipped -- it sits above the partial aggregate,
/// never below) and, when `projections` is set, a `ProjectionExec` whose expressions are all plain columns kept under
/// their own names (DataFusion < 43 dropped the predicate's columns with one between the aggregate and the filter; from
/// 43 on the same thing is the FilterExec's embedded projection).  The substituted node executes the SCAN directly and
/// reports the scan's partitioning, so the peeled RepartitionExec disappears with the nodes it fed.
fn peel(plan: &Arc<dyn ExecutionPlan>, projections: bool) -> Arc<dyn ExecutionPlan> {
    let mut cur = plan.clone();
    loop {
        let next = if let Some(c) = cur.as_any().downcast_ref::<CoalesceBatchesExec>() {
            c.input().clone()
        } else if let Some(r) = cur.as_any().downcast_ref::<RepartitionExec>() {
            match r.partitioning() {
                Partitioning::RoundRobinBatch(_) => r.input().clone(),
                _ => return cur,
            }
        } else if let Some(p) = cur.as_any().downcast_ref::<ProjectionExec>() {
            let plain = projections && p.expr().iter().all(|(e, alias)| col_name(e).map_or(false, |(n, _)| &n == alias));
            if !plain {
                return cur;
            }
            p.input().clone()
        } else {
            return cur;
        };
        cur = next;
    }
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
        // below the aggregate: [C] <- FilterExec <- [C] <- scan, or [C] <- scan when the only predicate was pushed down
        // into the scan ([C]: see `peel`)
        let below = peel(agg.input(), true);
        let matched = if let Some(filter) = below.as_any().downcast_ref::<FilterExec>() {
            // a FilterExec with an embedded projection only drops columns above it; the predicate is written against the
            // filter's INPUT (the scan's schema) and every matcher goes by column NAME, so it is accepted as it is
            let scan_plan = peel(filter.input(), false);
            if let Some(scan) = scan_plan.as_any().downcast_ref::<VCFScan>() {
                match_vcf_filter(agg, filter.predicate(), &scan_plan, scan.base_config(), None)
            } else if let Some(scan) = scan_plan.as_any().downcast_ref::<IndexedVCFScanner>() {
                match indexed_vcf_region(scan) {
                    Some(region) => match_vcf_filter(agg, filter.predicate(), &scan_plan, scan.base_config(), Some(region)),
                    None => None,
                }
            } else if ALIGNMENT_SCANS.contains(&scan_plan.name()) {
                match_alignment_filter(agg, filter.predicate(), &scan_plan)
            } else {
                None
            }
        } else if let Some(scan) = below.as_any().downcast_ref::<IndexedVCFScanner>() {
            match_indexed_vcf(agg, &below, scan)
        } else if let Some(scan) = below.as_any().downcast_ref::<IndexedBAMScan>() {
            match_indexed_bam(agg, &below, scan)
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
    //! To run on a machine with cargo + an MI355X.  One test per plan shape of the table above, each mirroring the
    //! reference's own plan-shape test (exon-core/src/datasources/vcf/table_provider.rs:571-611): build the session, plan
    //! the SQL, look for the node, run it and compare with the value the reference's slt files pin.
    use super::*;
    use datafusion::execution::session_state::SessionStateBuilder;
    use datafusion::prelude::SessionContext;
    use exon::ExonSession;

    fn session() -> std::result::Result<ExonSession, Box<dyn std::error::Error>> {
        let exon = ExonSession::new_exon()?; // Result<ExonSession>
        let state = SessionStateBuilder::new_from_existing(exon.session.state())
            .with_physical_optimizer_rule(Arc::new(GpuFilterAggRule::new(0)))
            .build();
        Ok(ExonSession::new(SessionContext::new_with_state(state))) // returns Self, not a Result (exon_context_ext.rs:108-112)
    }
    async fn plan_text(ctx: &ExonSession, sql: &str) -> std::result::Result<String, Box<dyn std::error::Error>> {
        let df = ctx.session.sql(sql).await?;
        let plan = ctx.session.state().create_physical_plan(df.logical_plan()).await?;
        Ok(datafusion::physical_plan::displayable(plan.as_ref()).indent(true).to_string())
    }
    async fn one_i64(ctx: &ExonSession, sql: &str) -> std::result::Result<i64, Box<dyn std::error::Error>> {
        let batches = ctx.session.sql(sql).await?.collect().await?;
        Ok(batches[0].column(0).as_any().downcast_ref::<arrow::array::Int64Array>().unwrap().value(0))
    }

    #[test]
    fn region_debug_parser() {
        let d = r#"IndexedVCFScanner { base_config: .., region: Region { name: "1", interval: Interval { start: Some(Position(9999921)), end: None } }, properties: .. }"#;
        assert_eq!(region_from_debug(d).as_deref(), Some("1:9999921"));
        let d = r#"region: Region { name: "chr1", interval: Interval { start: Some(Position(1)), end: Some(Position(12209145)) } }"#;
        assert_eq!(region_from_debug(d).as_deref(), Some("chr1:1-12209145"));
        let d = r#"region: Region { name: "1", interval: Interval { start: None, end: None } }"#;
        assert_eq!(region_from_debug(d).as_deref(), Some("1"));
        assert_eq!(region_from_debug("no region here"), None);
    }

    /// C4: `F <- VCFScan`
    #[tokio::test]
    async fn plan_shape_c4() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session.sql("SET exon.vcf_parse_info = true").await?;
        ctx.session.sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf'").await?;
        let sql = "SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info.\"MQ0F\" > -1 GROUP BY filter";
        let shown = plan_text(&ctx, sql).await?;
        assert!(shown.contains("GpuFilterAggExec") && shown.contains("AggregateExec: mode=Final"), "{shown}");
        let rows: usize = ctx.session.sql(sql).await?.collect().await?.iter().map(|b| b.num_rows()).sum();
        assert_eq!(rows, 1); // all 621 records carry the empty FILTER list
        Ok(())
    }

    /// C2: `F <- VCFScan` with the conjunction spelled out; 191 = slt/vcf-indexed-tests.slt:31-35
    #[tokio::test]
    async fn plan_shape_c2() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session.sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf'").await?;
        let sql = "SELECT COUNT(*) FROM v WHERE chrom = '1' AND pos >= 1";
        assert!(plan_text(&ctx, sql).await?.contains("GpuFilterAggExec"));
        assert_eq!(one_i64(&ctx, sql).await?, 191);
        Ok(())
    }

    /// C2r: `A <- IndexedVCFScanner`, no FilterExec (the shape of table_provider.rs:571-611)
    #[tokio::test]
    async fn plan_shape_c2_region_pushdown() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session
            .sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf.gz' OPTIONS (compression gzip)")
            .await?;
        let sql = "SELECT COUNT(*) FROM v WHERE vcf_region_filter('1', chrom)";
        let shown = plan_text(&ctx, sql).await?;
        assert!(shown.contains("GpuFilterAggExec") && shown.contains("IndexedVCFScanner") && !shown.contains("FilterExec"), "{shown}");
        assert_eq!(one_i64(&ctx, sql).await?, 191);
        Ok(())
    }

    /// C3 and C6 over `F <- BAMScan`; 61 rows in the fixture (slt/bam-select-tests.slt:56-59), 7 in chr1:1-12209145 (:16-19 of
    /// bam-indexed-select-tests.slt)
    #[tokio::test]
    async fn plan_shape_c3_c6() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session.sql("CREATE EXTERNAL TABLE b STORED AS BAM LOCATION 'test-data/datasources/bam/test.bam'").await?;
        let c3 = "SELECT reference, COUNT(*) FROM b WHERE flag & 1284 = 0 AND CAST(mapping_quality AS INT) >= 0 GROUP BY reference";
        assert!(plan_text(&ctx, c3).await?.contains("GpuFilterAggExec"));
        let c6 = "SELECT COUNT(*) FROM b WHERE reference = 'chr1' AND start <= 12209145 AND \"end\" >= 1";
        assert!(plan_text(&ctx, c6).await?.contains("GpuFilterAggExec"));
        assert_eq!(one_i64(&ctx, c6).await?, 7);
        Ok(())
    }

    /// The headline one-file C4 plan as DataFusion 44 emits it under `new_exon_config` (target_partitions = num_cpus,
    /// round-robin repartitioning on): `RepartitionExec(RoundRobinBatch)` between the FilterExec and the one-partition
    /// VCFScan (INTEGRATION.md section 3 has the expected EXPLAIN).  The rule must substitute anyway, and the
    /// substituted node must not keep the repartition below it.
    #[tokio::test]
    async fn plan_shape_c4_behind_round_robin_repartition() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session.sql("SET datafusion.execution.target_partitions = 8").await?;
        ctx.session.sql("SET exon.vcf_parse_info = true").await?;
        ctx.session.sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf'").await?;
        let sql = "SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info.\"MQ0F\" > -1 GROUP BY filter";
        let shown = plan_text(&ctx, sql).await?;
        assert!(shown.contains("GpuFilterAggExec"), "{shown}");
        let below = &shown[shown.find("GpuFilterAggExec").unwrap()..];
        assert!(!below.contains("RoundRobinBatch") && !below.contains("FilterExec") && below.contains("VCFScan"), "{shown}");
        let sql = "SELECT COUNT(*) FROM v WHERE chrom = '1' AND pos >= 1";
        assert!(plan_text(&ctx, sql).await?.contains("GpuFilterAggExec"));
        assert_eq!(one_i64(&ctx, sql).await?, 191);
        Ok(())
    }

    /// empty / non-positive intervals are left to DataFusion (`pos <= 0` used to become the region text `1:1-0`)
    #[tokio::test]
    async fn empty_intervals_are_not_substituted() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        ctx.session.sql("CREATE EXTERNAL TABLE v STORED AS VCF LOCATION 'test-data/datasources/vcf/index.vcf'").await?;
        for sql in ["SELECT COUNT(*) FROM v WHERE chrom = '1' AND pos <= 0", "SELECT COUNT(*) FROM v WHERE chrom = '1' AND pos >= 10 AND pos <= 5"] {
            assert!(!plan_text(&ctx, sql).await?.contains("GpuFilterAggExec"), "{sql}");
            assert_eq!(one_i64(&ctx, sql).await?, 0);
        }
        Ok(())
    }

    /// C5: the table function; 2 reads in the fixture (slt/fastq-scan-test.slt:51-55)
    #[tokio::test]
    async fn fastq_histogram_udtf() -> std::result::Result<(), Box<dyn std::error::Error>> {
        let ctx = session()?;
        crate::udtf::register(&ctx.session, 0);
        let sql = "SELECT SUM(count) FROM fastq_quality_histogram('test-data/datasources/fastq/test.fastq') WHERE position = 0";
        assert_eq!(one_i64(&ctx, sql).await?, 2);
        Ok(())
    }
}
