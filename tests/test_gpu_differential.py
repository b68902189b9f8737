"""Differential parity: the whole -part1 -part2 pipeline on the GPU against the CPU oracle on synthetic maps the
golden fixtures do not cover - a few very large scaffolds (windows of hundreds of bins), mostly one-bin scaffolds
(orientation ties, long insertion queues with host-decided steps), quantised sparse contacts (exact ties in the
clustering, the rank order and the ordering scores) and other brute-force / scan window sizes.  Every output file
must be identical (the oracle is pinned to the reference by tests/golden/, see oracle/hic_oracle.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name, bins, chromosomes, mean scaffold bins, quantise, nScaffolds, scanScaffolds, psig
    ("huge-scaffolds", 420, 3, 45.0, None, 6, 5, .05),
    ("one-bin-scaffolds", 260, 3, 1.6, None, 6, 5, .05),
    ("quantised-sparse", 360, 4, 7.0, (0.35, 2), 6, 5, .05),
    ("small-windows", 300, 3, 9.0, None, 4, 3, .01),
    ("k-equals-scan", 280, 2, 11.0, (0.2, 3), 5, 5, .05),
    # a few more draws of the generator's knobs
    ("draw-a", 330, 4, 4.0, (0.5, 1), 5, 4, .05),
    ("draw-b", 250, 2, 14.0, None, 6, 4, .001),
    ("draw-c", 310, 5, 5.5, (0.1, 2), 4, 4, .05),
    ("draw-d", 200, 1, 3.0, None, 6, 5, .05),
]

# Part 2 alone on the planted chromosomes (the cut scan over-segments maps this small, which would leave no
# chromosome with more scaffolds than the windows are wide)
P2_CASES = [
    # name, bins, chromosomes, mean scaffold bins, quantise, nScaffolds, scanScaffolds
    ("two-scaffold-windows", 180, 2, 6.0, None, 3, 2),
    ("six-scaffold-scan-windows", 150, 1, 9.0, None, 6, 6),
    ("seven-scaffold-brute-force", 120, 1, 12.0, (0.2, 2), 7, 4),
]


def _contacts(lay, seed, quantise):
    from hic_genome_assembler_amd import synth
    c = synth.dense_contacts(lay, seed=seed, sinkhorn_iters=8)
    if quantise is not None:
        frac, decimals = quantise
        c = np.round(c, decimals)                           # equal values everywhere
        cut = np.quantile(c, frac)
        c[c <= cut] = 0.0                                   # ... and many exact zeros
        c = 0.5 * (c + c.T)
        np.fill_diagonal(c, np.maximum(np.diag(c), 1.0))    # no empty rows from the sparsification itself
    return np.ascontiguousarray(c)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_pipeline_matches_oracle(case, tmp_path):
    import hic_oracle as orc
    from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1, synth
    name, n, n_chrom, mean_scaf, quantise, n_scaffolds, scan_scaffolds, psig = case
    seed = 100 + len(name)
    lay = synth.make_layout(n, seed=seed, n_chrom=n_chrom, mean_scaffold_bins=mean_scaf)
    c = _contacts(lay, seed, quantise)
    paths = synth.write_hicpro(str(tmp_path / "in"), lay, c, "d")
    outs = {}
    for who in ("oracle", "gpu"):
        out = tmp_path / who
        out.mkdir()
        f = lambda k: str(out / k)  # noqa: E731
        if who == "oracle":
            orc.run_part1(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                          paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("binGroups.txt"), f("assessment.txt"),
                          f("chromosomeGroups.txt"), min_size=5, modularity=0.0, psig=psig)
            orc.run_part2(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                          f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), f("plotOrder.txt"),
                          n_scaffolds=n_scaffolds, scan_scaffolds=scan_scaffolds)
        else:
            p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                           paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), False, False, f("binGroups.txt"),
                           f("assessment.txt"), f("chromosomeGroups.txt"), True, False, 5, 0.0, 1, psig, 5, 5, lay.resolution)
            p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                           f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), False, False, False, "t", f("plotOrder.txt"),
                           n_scaffolds, scan_scaffolds, lay.resolution)
        outs[who] = {k: open(f(k)).read() for k in ("dendrogramOrder.txt", "binGroups.txt", "assessment.txt",
                                                    "chromosomeGroups.txt", "chromosomeOrders.txt", "plotOrder.txt")}
    for k in outs["oracle"]:
        assert outs["gpu"][k] == outs["oracle"][k], (name, k)
    assert len(outs["gpu"]["plotOrder.txt"]) > 0


@pytest.mark.parametrize("case", P2_CASES, ids=[c[0] for c in P2_CASES])
def test_part2_matches_oracle_on_planted_groups(case, tmp_path):
    import hic_oracle as orc
    from hic_genome_assembler_amd import orderGenome as p2, synth
    name, n, n_chrom, mean_scaf, quantise, n_scaffolds, scan_scaffolds = case
    seed = 200 + len(name)
    lay = synth.make_layout(n, seed=seed, n_chrom=n_chrom, mean_scaffold_bins=mean_scaf)
    c = _contacts(lay, seed, quantise)
    paths = synth.write_hicpro(str(tmp_path / "in"), lay, c, "d")
    groups = tmp_path / "chromosomeGroups.txt"
    with open(groups, "w") as fh:
        for g in range(n_chrom):
            fh.write("### Chromosome group %d ###\n" % (g + 1))
            for k in np.flatnonzero(lay.chrom_of_bin == g):
                fh.write("%d\t%s\n" % (lay.bin_ids[k], lay.scaffold_names[lay.scaffold_of_bin[k]]))
    outs = {}
    for who in ("oracle", "gpu"):
        out = tmp_path / who
        out.mkdir()
        f = lambda k: str(out / k)  # noqa: E731
        if who == "oracle":
            orc.run_part2(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"], str(groups),
                          f("chromosomeOrders.txt"), f("plotOrder.txt"), n_scaffolds=n_scaffolds, scan_scaffolds=scan_scaffolds)
        else:
            p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"], str(groups),
                           f("chromosomeOrders.txt"), False, False, False, "t", f("plotOrder.txt"), n_scaffolds,
                           scan_scaffolds, lay.resolution)
        outs[who] = {k: open(f(k)).read() for k in ("chromosomeOrders.txt", "plotOrder.txt")}
    assert outs["gpu"] == outs["oracle"], name


def test_part1_matches_oracle_at_6000_bins(tmp_path):
    """Part 1 at three times the size of the largest reference-generated fixture: UPGMA leaf order, rank-order cut
    scan with its M-shrinking rescans, noisy-cut filter and scaffold voting - four files, byte for byte (the contacts
    are handed over in memory; the text loaders have their own tests)."""
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, scaffoldToChromosomes as p1, synth
    from hic_genome_assembler_amd.hostio import Bin
    n = 6000
    lay = synth.make_layout(n, seed=9)
    c = synth.dense_contacts(lay, seed=9, sinkhorn_iters=6)
    sizes = tmp_path / "sizes.txt"
    sizes.write_text("".join("%s\t%d\n" % (nm, sz) for nm, sz in zip(lay.scaffold_names, lay.scaffold_sizes_bp)))
    mk = lambda cls: [cls(int(lay.bin_ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], int(lay.start[k]), int(lay.stop[k]),
                          1.0, 0.0) for k in range(n)]  # noqa: E731
    names = ("dendrogramOrder.txt", "binGroups.txt", "assessment.txt", "chromosomeGroups.txt")
    for who in ("oracle", "gpu"):
        (tmp_path / who).mkdir()
    fo = [str(tmp_path / "oracle" / k) for k in names]
    fg = [str(tmp_path / "gpu" / k) for k in names]
    cuts_o = orc.run_part1(None, None, None, str(sizes), *fo, min_size=5, modularity=0.0, psig=.05, preloaded=(c, mk(orc.Bin)))
    with _lib.Context(0) as ctx:
        ctx.set_contacts(c)
        cuts_g = p1.runResident(p1.DeviceMatrix(ctx), mk(Bin), str(sizes), *fg, 5, 0.0, .05)
    assert list(cuts_g) == list(cuts_o) and len(cuts_g) >= 8
    for a, b in zip(fg, fo):
        assert open(a).read() == open(b).read(), a


_RESIDENT_RUN = r"""
import contextlib, io, os, sys
sys.path.insert(0, sys.argv[1])
import torch
from hic_genome_assembler_amd import _lib, synth, orderGenome as p2, scaffoldToChromosomes as p1
from hic_genome_assembler_amd.hostio import Bin
n, out = int(sys.argv[2]), sys.argv[3]
lay = synth.make_layout(n, seed=4)
dev = torch.device("cuda", 0)
c = synth.dense_contacts_torch(lay, dev, seed=4, sinkhorn_iters=8)
torch.cuda.synchronize()
sizes = os.path.join(out, "sizes.txt")
with open(sizes, "w") as fh:
    fh.write("".join("%s\t%d\n" % (nm, sz) for nm, sz in zip(lay.scaffold_names, lay.scaffold_sizes_bp)))
bins = [Bin(int(lay.bin_ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], int(lay.start[k]), int(lay.stop[k]), 1.0, 0.)
        for k in range(n)]
f = lambda k: os.path.join(out, k)
ctx = _lib.Context(0)
ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
dm = p1.DeviceMatrix(ctx)
with contextlib.redirect_stdout(io.StringIO()):
    p1.runResident(dm, bins, sizes, f("dendrogramOrder.txt"), f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                   5, 0.0, .05)
    p2.runResident(p2.GenomeMatrix(ctx), dm.kept_bins, f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), f("plotOrder.txt"),
                   6, 5, lay.resolution)
"""


def test_fast_paths_agree_with_the_earlier_ones_at_scale(tmp_path):
    """An 8,000-bin map (chromosomes of ~900 bins, ~70 scaffolds - far beyond what the Python oracle can order in a
    test) through the default paths - lock-step device-decided insertion, placement tables, register-blocked sort -
    and through the earlier implementations of the same stages (host-decided insertion, per-candidate window kernels,
    LDS sort network, one queue per chromosome), which the golden fixtures validated first.  Same files."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(_RESIDENT_RUN)
    variants = {"default": {},
                "earlier": {"HICMI_P2_HOST_INSERT": "1", "HICMI_P2_WINDOW_DIRECT": "1", "HICMI_SORT_LDS": "1",
                            "HICMI_PART2_LOCKSTEP": "0"}}
    texts = {}
    for name, env in variants.items():
        out = tmp_path / name
        out.mkdir()
        res = subprocess.run([sys.executable, str(script), root, "8000", str(out)], env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        texts[name] = {k: open(out / k).read() for k in ("dendrogramOrder.txt", "binGroups.txt", "chromosomeGroups.txt",
                                                         "chromosomeOrders.txt", "plotOrder.txt")}
    assert texts["default"] == texts["earlier"]
    assert texts["default"]["chromosomeOrders.txt"].count("\n") > 400
