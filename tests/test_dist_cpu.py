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


def _run_part2_sharded(rank, world, name, tmp):
    """One map over ``world`` ranks: Part 1 replicated, Part 2's chromosomes dealt to the ranks (orderGenome's
    ``shard``), the ordered lists all-gathered, rank 0 writes the files."""
    import contextlib
    import io
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import golden_cases as gc
    from fake_context import OracleContext
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, orderGenome as p2, scaffoldToChromosomes as p1
    from hic_genome_assembler_amd.hostio import initiateLoci
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
        chroms = p2.readChromsFromFile(f("chromosomeGroups.txt"))
        mine = p2.chromosomesOfRank(chroms, rank, world)
        binDict = p2.readGroupingsToValidBins(f("chromosomeGroups.txt"))
        binList = initiateLoci(paths["hicProBedFile"], paths["hicProBiasFile"], binID_dict=binDict)
        adj = p2.buildAdjacencyMatrix(paths["hicProMatrixFile"], binList)
        seen = []
        inner = p2.orderChromosome

        def spy(group, *a, **k):
            seen.append(chroms.index(group))
            return inner(group, *a, **k)
        spy.last_cost = None
        p2.orderChromosome = spy
        try:
            ordered = p2.runResident(adj, binList, f("chromosomeGroups.txt"), f("chromosomeOrders.txt"),
                                     f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000,
                                     shard=(rank, world))
        finally:
            p2.orderChromosome = inner
    assert sorted(seen) == mine, (seen, mine)                      # this rank ordered its own deal only
    assert len(ordered) == len(chroms)                            # ... and still holds the whole genome order
    wrote = os.path.exists(f("chromosomeOrders.txt"))
    assert wrote == (rank == 0)
    return {fn: open(f(fn)).read() for fn in ("chromosomeOrders.txt", "plotOrder.txt")} if wrote else None, mine


def _shard_worker(rank, world, port, tmp_root, name, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path[:0] = [ROOT]
    from hic_genome_assembler_amd import dist
    dist.init("gloo")
    tmp = os.path.join(tmp_root, "shard_r%d" % rank)
    os.makedirs(tmp)
    files, mine = _run_part2_sharded(rank, world, name, tmp)
    deals = dist.gather_results({rank: mine})
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "sharded.json"), "w") as fh:
            json.dump({"files": files, "deals": {str(k): v for k, v in deals.items()}}, fh)
    dist.barrier()


@pytest.mark.parametrize("name", ["n600", "n300_edges"])
def test_one_map_part2_sharded_over_two_ranks(tmp_path, name):
    import json
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import golden_cases as gc
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path), name, str(tmp_path)), nprocs=2, join=True)
    with open(tmp_path / "sharded.json") as fh:
        got = json.load(fh)
    for fn in ("chromosomeOrders.txt", "plotOrder.txt"):
        assert got["files"][fn] == gc.golden_text(name, fn), fn
    a, b = got["deals"]["0"], got["deals"]["1"]
    assert not set(a) & set(b) and len(a) + len(b) > 0
    assert sorted(a + b) == list(range(len(a) + len(b)))


def test_chromosome_deal_is_balanced_and_complete():
    sys.path[:0] = [ROOT]
    from hic_genome_assembler_amd.orderGenome import chromosomesOfRank
    chroms = [[0] * k for k in (50, 400, 120, 120, 300, 10, 80)]
    for world in (1, 2, 3, 8, 16):
        deals = [chromosomesOfRank(chroms, r, world) for r in range(world)]
        assert sorted(i for d in deals for i in d) == list(range(len(chroms)))
    two = [chromosomesOfRank(chroms, r, 2) for r in range(2)]
    assert two[0] == [1] or 1 in two[0]                           # the largest goes to rank 0
    load = [sum(len(chroms[i]) ** 2 for i in d) for d in two]
    assert max(load) <= 1.3 * min(load)


# ---- one map, Part 1's row-independent stages sharded over the ranks (first bullet) ----------------------
def _run_part1_sharded(rank, world, name, tmp):
    import contextlib
    import io
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import golden_cases as gc
    from fake_context import OracleContext
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, scaffoldToChromosomes as p1
    seen = {"scans": 0, "foreign_nonzero": 0, "owned_rows": 0}

    class Spy(OracleContext):
        def cut_scan(self, start, M, psig, want_x=False):
            sig = OracleContext.cut_scan(self, start, M, psig, want_x)
            rows = np.arange(start, start + len(sig))
            seen["scans"] += 1
            seen["foreign_nonzero"] += int(np.count_nonzero(sig[rows % world != rank]))
            seen["owned_rows"] += int(np.count_nonzero(rows % world == rank))
            return sig

        def filter_scan(self, start, c, n_rows, M, psig, want_x=False):
            sig = OracleContext.filter_scan(self, start, c, n_rows, M, psig, want_x)
            rows = np.arange(start, start + len(sig))
            seen["scans"] += 1
            seen["foreign_nonzero"] += int(np.count_nonzero(sig[rows % world != rank]))
            return sig
    _lib.Context = Spy
    _lib.hypergeom_sf = lambda x, M, n, N: float(orc.hyper_geom(x, M, n, N))
    spec = gc.load_case(name)[0]
    paths = gc.write_case_files(name, tmp)
    f = lambda k: os.path.join(tmp, k)  # noqa: E731
    with contextlib.redirect_stdout(io.StringIO()):
        p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("a.png"), f("b.png"),
                       f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                       True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000, shard=(rank, world))
    files = {fn: open(f(fn)).read() for fn in ("dendrogramOrder.txt", "binGroups.txt", "assessment.txt", "chromosomeGroups.txt")}
    return files, seen


def _part1_shard_worker(rank, world, port, tmp_root, name, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path[:0] = [ROOT]
    from hic_genome_assembler_amd import dist
    dist.init("gloo")
    tmp = os.path.join(tmp_root, "p1_r%d" % rank)
    os.makedirs(tmp)
    files, seen = _run_part1_sharded(rank, world, name, tmp)
    everything = dist.gather_results({rank: {"files": files, "seen": seen}})
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "part1_sharded.json"), "w") as fh:
            json.dump({str(k): v for k, v in everything.items()}, fh)
    dist.barrier()


@pytest.mark.parametrize("name,world", [("n600", 2), ("n300_edges", 3)])
def test_one_map_part1_rows_sharded_over_the_ranks(tmp_path, name, world):
    """Every rank sums, sorts and scans only the rows r == rank (mod world) (here: the oracle-backed context answers
    0 for everybody else's rows, as libhicmi does after hicmi_set_row_shard); one all-gather per vector gives every
    rank the full flags, and all ranks write the reference's four Part 1 files."""
    import json
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import golden_cases as gc
    port = 29600 + (os.getpid() % 90) + world
    mp.spawn(_part1_shard_worker, args=(world, port, str(tmp_path), name, str(tmp_path)), nprocs=world, join=True)
    with open(tmp_path / "part1_sharded.json") as fh:
        got = json.load(fh)
    assert sorted(got) == [str(r) for r in range(world)]
    for r in range(world):
        for fn, text in got[str(r)]["files"].items():
            assert text == gc.golden_text(name, fn), (r, fn)
        seen = got[str(r)]["seen"]
        assert seen["scans"] > 10 and seen["foreign_nonzero"] == 0 and seen["owned_rows"] > 0


def test_gather_owned_single_process_is_identity():
    sys.path[:0] = [ROOT]
    import numpy as np
    from hic_genome_assembler_amd import dist
    a = np.arange(7, dtype=np.int32)
    assert dist.gather_owned(a, 3, 0, 1) is not None and np.array_equal(dist.gather_owned(a, 3, 0, 1), a)
