"""world_size-2 gloo run of the multi-rank path on the CPU: maps are dealt to ranks, each rank runs
the product's host flow (on the oracle-backed test context), timings are MAX-reduced and results
gathered - and they must equal the single-process results."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_map(name, tmp):
    import contextlib
    import io
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import golden_cases as gc
    from fake_context import OracleContext
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, orderGenome as p2, scaffoldToChromosomes as p1
    _lib.Context = OracleContext
    _lib.hypergeom_sf = lambda x, M, n, N: float(orc.hyper_geom(x, M, n, N))
    spec = gc.load_case(name)[0]
    paths = gc.write_case_files(name, tmp)
    f = lambda k: os.path.join(tmp, k)  # noqa: E731
    with contextlib.redirect_stdout(io.StringIO()):
        p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("a.png"), f("b.png"),
                       f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                       True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000)
        p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), tmp, "s", f("g.png"), "t",
                       f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000)
    return {fn: open(f(fn)).read() for fn in gc.OUTPUT_FILES}


def _worker(rank, world, port, tmp_root, names, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path[:0] = [ROOT]
    import time
    from hic_genome_assembler_amd import dist
    r, w = dist.init("gloo")
    assert (r, w) == (rank, world)
    mine = dist.units_of_rank(len(names), rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    local = {}
    for u in mine:
        tmp = os.path.join(tmp_root, "r%d_u%d" % (rank, u))
        os.makedirs(tmp)
        local[u] = _run_map(names[u], tmp)
    elapsed = time.perf_counter() - t0
    slowest = dist.max_over_ranks(elapsed)
    assert slowest >= elapsed
    allres = dist.gather_results(local)
    assert sorted(allres) == list(range(len(names)))
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "gathered.json"), "w") as fh:
            json.dump({str(k): v for k, v in allres.items()}, fh)
    dist.barrier()


def test_two_ranks_gloo(tmp_path):
    import json
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import golden_cases as gc
    names = ["n160", "n300_edges", "n160"]
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(2, port, str(tmp_path), names, str(tmp_path)), nprocs=2, join=True)
    with open(tmp_path / "gathered.json") as fh:
        got = json.load(fh)
    for u, name in enumerate(names):
        for fn in gc.OUTPUT_FILES:
            assert got[str(u)][fn] == gc.golden_text(name, fn), (name, fn)


def test_units_are_dealt_round_robin():
    sys.path[:0] = [ROOT]
    from hic_genome_assembler_amd import dist
    assert dist.units_of_rank(5, 0, 2) == [0, 2, 4] and dist.units_of_rank(5, 1, 2) == [1, 3]
    assert sorted(dist.units_of_rank(7, 0, 3) + dist.units_of_rank(7, 1, 3) + dist.units_of_rank(7, 2, 3)) == list(range(7))
