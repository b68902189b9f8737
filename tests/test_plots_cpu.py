"""Plot emission, host side: the figure writer with NumPy doing the reductions, and the
pipeline writing every file of the reference's plot layout (device calls answered by the oracle)."""
import os

import numpy as np
import pytest

import golden_cases as gc


@pytest.fixture()
def plots_on(monkeypatch):
    monkeypatch.delenv("HICMI_NO_PLOTS", raising=False)


def test_host_image_block_means_and_png(tmp_path, plots_on):
    from hic_genome_assembler_amd import plotContactMaps as pm
    rng = np.random.default_rng(0)
    a = rng.random((37, 37))
    img = pm._HostImage(a)
    got = img.pixels(5)
    edges = (np.arange(6) * 37) // 5
    want = np.array([[a[edges[r]:edges[r + 1], edges[c]:edges[c + 1]].mean() for c in range(5)] for r in range(5)])
    assert np.allclose(got, want, rtol=1e-13, atol=0)
    assert img.pixels(64) is img.a                                     # never upsampled
    out = tmp_path / "small.png"
    pm.plotContactMap(a, resolution=100000, highlightChroms=[10, 20], wInches=4, hInches=4, savePlot=str(out),
                      title="t", titleSuffix="_s")
    assert out.exists() and out.stat().st_size > 1000
    assert pm.figure_pixels(16000, 32, 32) == 3200 and pm.figure_pixels(160, 24, 24) == 160


def test_pipeline_writes_the_reference_plot_layout(tmp_path, plots_on, monkeypatch):
    from fake_context import OracleContext
    from hic_genome_assembler_amd import _lib, orderGenome as p2, plotContactMaps as pm, scaffoldToChromosomes as p1
    monkeypatch.setattr(_lib, "Context", OracleContext)
    monkeypatch.setattr(pm, "MAX_PIXELS", 200)                         # keep the figures small
    name = "n160"
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    out = tmp_path / "out"; plots = tmp_path / "plots"
    out.mkdir(); plots.mkdir()
    f = lambda k: str(out / k)  # noqa: E731
    p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"], paths["hicProScaffSizeFile"],
                   f("dendrogramOrder.txt"), str(plots / "avgCluster.png"), str(plots / "avgCluster_outlined.png"),
                   f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"), True, False, spec["min_size"], 0.0,
                   1, spec["psig"], 5, 5, lay.resolution)
    p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"], f("chromosomeGroups.txt"),
                   f("chromosomeOrders.txt"), str(plots), "_suffix", str(plots / "fullGenome.png"), "Genome",
                   f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], lay.resolution)
    n_groups = sum(1 for l in open(f("chromosomeGroups.txt")) if l.startswith("#"))
    want = ["avgCluster.png", "avgCluster_outlined.png", "fullGenome.png"] + ["Chr_%d.png" % (i + 1) for i in range(n_groups)]
    for fn in want:
        assert (plots / fn).exists() and (plots / fn).stat().st_size > 1000, fn
    # plotting must not disturb the text outputs
    for fn in gc.OUTPUT_FILES:
        with open(f(fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn
