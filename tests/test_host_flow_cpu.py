"""The product's host control flow (Part 1 cut/filter loops, Part 2 search, writers) driven on the
CPU through tests/fake_context.py, against the reference-generated fixtures."""
import os

import numpy as np
import pytest

import golden_cases as gc
from fake_context import OracleContext

CASES = ["n160", "n300_edges", "n600", "n400_default"]


@pytest.fixture()
def fake_gpu(monkeypatch):
    from hic_genome_assembler_amd import _lib
    monkeypatch.setattr(_lib, "Context", OracleContext)
    monkeypatch.setattr(_lib, "hypergeom_sf", lambda x, M, n, N: float(__import__("hic_oracle").hyper_geom(x, M, n, N)))
    return _lib


@pytest.mark.parametrize("name", CASES)
def test_host_flow_reproduces_reference_files(fake_gpu, name, tmp_path):
    from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    out = str(tmp_path)
    f = lambda k: os.path.join(out, k)  # noqa: E731
    p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                   paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("a.png"), f("b.png"),
                   f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                   True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000)
    scores = []
    p2.SCORE_HOOK = lambda fast: scores.extend(float(v) for v in fast)
    try:
        p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), out, "synthetic", f("g.png"),
                       "synthetic genome", f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000)
    finally:
        p2.SCORE_HOOK = None
    for fn in gc.OUTPUT_FILES:
        with open(f(fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn
    assert len(scores) == len(gold["costs"])
    assert np.allclose(scores, gold["costs"], rtol=1e-12, atol=0)
