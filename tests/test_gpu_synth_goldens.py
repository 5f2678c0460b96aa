"""The HIP kernels against ORACLE-INDEPENDENT golden files (tests/golden/synth/*.npz, built by
tests/golden/make_synth_goldens.py from numpy + pyarrow.compute only: SURVEY section 8(c) "golden vectors to commit (ii)").
The inputs are regenerated from the seeds (their SHA-256 is in the file, so a different numpy stream is reported as such and
not as a kernel bug); the expectations are READ FROM THE FILES -- oracle/ is not imported here.  Bit-exact for counts and
histograms; the f64 sums are sums of eighths below 2^53, so they are compared for equality too."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("make_synth_goldens", os.path.join(ROOT, "tests", "golden", "make_synth_goldens.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


def load(name):
    """(inputs regenerated from the seed and verified against the stored hashes, expectations from the file)"""
    f = np.load(os.path.join(G.OUT, name + ".npz"))
    c = G.CASES[name]
    gen = {"c2": lambda: G.gen_c2(c["seed"], c["n"]), "c3": lambda: G.gen_c3(c["seed"], c["n"], c["n_refs"]),
           "c4": lambda: G.gen_c4(c["seed"], c["n"], c["n_groups"], c["specials"]),
           "c5": lambda: G.gen_c5(c["seed"], c["n"], c["lo"], c["hi"]), "c6": lambda: G.gen_c6(c["seed"], c["n"])}[c["kind"]]
    d = gen()
    for k, v in d.items():
        assert G.sha(v) == str(f["sha256_" + k]), f"{name}: numpy regenerated a different `{k}` than the golden file was built from"
    return c, d, {k: f[k] for k in f.files if not k.startswith("sha256_")}


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_golden_files_are_what_the_script_builds(name):
    """CPU: the committed files equal a fresh run of the builder (numpy / pyarrow only)."""
    c, d, want = load(name)
    _, again = G.build(name)
    assert again.keys() == want.keys()
    for k in want:
        assert np.array_equal(again[k], want[k]), (name, k)


def _dev(ctx, a):
    return ctx.to_device(a)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(G.CASES) if G.CASES[n]["kind"] == "c2"])
def test_region_count_against_golden(ctx, name):
    c, d, want = load(name)
    n = c["n"]
    out = ctx.zeros(np.int64, 1)
    cid, a, b = c["region"]
    ctx.region_count(_dev(ctx, d["chrom"]), _dev(ctx, d["pos"]), n, cid, a, b, out, chrom_valid=_dev(ctx, G.bitmap(d["cvalid"])),
                     pos_valid=_dev(ctx, G.bitmap(d["pvalid"])))
    ctx.sync()
    assert out.to_host().tolist() == want["count"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(G.CASES) if G.CASES[n]["kind"] == "c3"])
def test_flag_mapq_group_count_against_golden(ctx, name):
    c, d, want = load(name)
    n, R = c["n"], c["n_refs"]
    out = ctx.zeros(np.int64, R + 1)
    mapq = np.concatenate([d["mapq"], np.zeros(64, np.uint8)])
    ctx.flag_mapq_group_count(_dev(ctx, d["flag"]), _dev(ctx, mapq), _dev(ctx, G.bitmap(d["mvalid"])), _dev(ctx, d["ref"]),
                              _dev(ctx, G.bitmap(d["rvalid"])), n, c["mask"], c["value"], c["qmin"], R, out,
                              flag_valid=_dev(ctx, G.bitmap(d["fvalid"])))
    ctx.sync()
    assert np.array_equal(out.to_host(), want["counts"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(G.CASES) if G.CASES[n]["kind"] == "c4"])
def test_cmp_avg_by_group_against_golden(ctx, name):
    c, d, want = load(name)
    n, NG = c["n"], c["n_groups"]
    dc, ds = ctx.zeros(np.int64, 2 * NG), ctx.zeros(np.float64, NG)
    ctx.cmp_avg_by_group(_dev(ctx, d["af"]), _dev(ctx, G.bitmap(d["avalid"])), _dev(ctx, d["qual"]), _dev(ctx, G.bitmap(d["qvalid"])),
                         _dev(ctx, d["gid"]), n, c["thr"], c["op"], NG, dc, ds)
    ctx.sync()
    got = dc.to_host()
    assert np.array_equal(got[:NG], want["count_y"]) and np.array_equal(got[NG:], want["count_rows"])
    assert np.array_equal(ds.to_host(), want["sum_y"])  # sums of eighths: exact in f64 whatever the order


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(G.CASES) if G.CASES[n]["kind"] == "c5"])
def test_qual_pos_hist_against_golden(ctx, name):
    c, d, want = load(name)
    lmax = c["lmax"]
    out = ctx.zeros(np.int64, lmax * 256)
    data = np.concatenate([d["data"], np.zeros(64, np.uint8)])
    ctx.qual_pos_hist(_dev(ctx, d["off"]), _dev(ctx, data), c["n"], lmax, out)
    ctx.sync()
    assert np.array_equal(out.to_host(), want["hist"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(G.CASES) if G.CASES[n]["kind"] == "c6"])
def test_overlap_and_within_count_against_golden(ctx, name):
    c, d, want = load(name)
    n = c["n"]
    out = ctx.zeros(np.int64, 1)
    rid, a, b = c["region"]
    args = (_dev(ctx, d["ref"]), _dev(ctx, G.bitmap(d["rvalid"])), _dev(ctx, d["start"]), _dev(ctx, G.bitmap(d["svalid"])),
            _dev(ctx, d["end"]), _dev(ctx, G.bitmap(d["evalid"])), n, rid, a, b, out)
    (ctx.within_count if c["strict"] else ctx.overlap_count)(*args)
    ctx.sync()
    assert out.to_host().tolist() == want["count"].tolist()
