"""CPU-only tests of the host side: C-ABI surface, config parser, enumeration order, loaders."""
import os
import re
import subprocess

import numpy as np
import pytest

import golden_cases as gc
import hic_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from hic_genome_assembler_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    with open(os.path.join(ROOT, "include", "hicmi.h")) as fh:
        text = fh.read()
    declared = set(re.findall(r"\b(hicmi_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (hicmi_[a-z0-9_]+)", out))
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(lib.SIGNATURES), sorted(declared ^ set(lib.SIGNATURES))
    assert lib.load().hicmi_abi_version() == 1


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.HicmiError):
        lib.Context(0)


def test_host_hypergeom_matches_scipy(lib):
    from scipy.stats import hypergeom
    assert lib.hypergeom_sf(8, 600, 50, 50) == pytest.approx(0.045750386176022624, rel=1e-12)
    assert lib.hypergeom_sf(7, 600, 50, 50) == pytest.approx(0.11023306317208498, rel=1e-12)
    rng = np.random.default_rng(1)
    for _ in range(3000):
        M = int(rng.integers(1, 40000)); L = int(rng.integers(1, M + 1))
        mean = L * L / M
        sd = max(1.0, (mean * (1 - L / M) * (M - L) / max(M - 1, 1)) ** 0.5)
        x = int(round(mean + rng.normal() * 2.5 * sd))
        mine, ref = lib.hypergeom_sf(x, M, L, L), float(hypergeom.sf(x - 1, M, L, L))
        assert (mine < .05) == (ref < .05) and (mine >= .05) == (ref >= .05)
        if ref > 1e-290:
            assert abs(mine - ref) <= 1e-10 * ref
    # argument checking as SciPy: invalid -> NaN, below support -> 1, above -> 0
    assert np.isnan(lib.hypergeom_sf(1, 5, 6, 2)) and np.isnan(lib.hypergeom_sf(1, -3, 0, 0))
    assert lib.hypergeom_sf(0, 10, 4, 4) == 1.0 and lib.hypergeom_sf(5, 10, 4, 4) == 0.0


def test_early_exit_hypergeom_decision_equals_the_full_sum(lib):
    """The scan kernels only need `hyper_geom(...) < psig` and stop the tail sum as soon as that is settled (hyper.h:
    hypergeom_decide, here in its host build): the decision must be that of the full sum and of SciPy - for draws on both
    sides of the mode, for the two shapes the scans use (n = N = L; the filter's scalar tests n != N) and for thresholds
    from tiny to almost 1."""
    from scipy.stats import hypergeom
    rng = np.random.default_rng(5)
    n_near = 0
    for it in range(6000):
        M = int(rng.integers(1, 66000))
        if it % 2:
            n = N = int(rng.integers(1, M + 1))
        else:
            n = int(rng.integers(0, M + 1)); N = int(rng.integers(0, M + 1))
        mean = n * N / M
        sd = max(1.0, (mean * (1 - n / M) * (M - N) / max(M - 1, 1)) ** 0.5)
        x = int(round(mean + rng.normal() * 2.5 * sd))
        full = lib.hypergeom_sf(x, M, n, N)
        ref = float(hypergeom.sf(x - 1, M, n, N))
        for psig in (.05, 1e-6, .5, .95, 1e-300):
            dec = lib.hypergeom_decide(x, M, n, N, psig)
            assert dec == (1 if full < psig else 0), (x, M, n, N, psig, full)
            if abs(ref - psig) > 1e-9 * psig:
                assert dec == (1 if ref < psig else 0), (x, M, n, N, psig, ref)
            n_near += abs(ref - psig) < .3 * psig
    assert n_near > 300                                       # (the thresholds are really being approached)
    assert lib.hypergeom_decide(1, 5, 6, 2, .05) == -1 and lib.hypergeom_decide(0, 10, 4, 4, .05) == 0
    assert lib.hypergeom_decide(5, 10, 4, 4, .05) == 1


def test_linkage_helpers_match_oracle(lib):
    import ctypes
    L = lib.load()
    rng = np.random.default_rng(2)
    for n in (2, 3, 50, 400):
        d = rng.random((n, n)) + 1
        zraw = orc.nn_chain_raw(d)
        z_o = orc.label_linkage(zraw, n)
        leaves_o = orc.leaf_order(z_o, n)
        z = np.zeros_like(zraw)
        assert L.hicmi_label_linkage(zraw.ctypes.data_as(ctypes.c_void_p), n, z.ctypes.data_as(ctypes.c_void_p)) == 0
        leaves = np.zeros(n, np.int32)
        assert L.hicmi_leaf_order(z.ctypes.data_as(ctypes.c_void_p), n, leaves.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(z, z_o) and np.array_equal(leaves, leaves_o)


def test_enumeration_order_matches_oracle():
    from hic_genome_assembler_amd import orderGenome as p2
    for k in range(1, 7):
        names = ["s%d" % i for i in range(k)]
        assert p2.removeReverseDuplicates(p2.permutations(list(names), [], 0)) == \
            orc.remove_reverse_duplicates(orc.swap_permutations(names))
        assert p2.plusMinusPerms(names) == orc.plus_minus_perms(k)
    assert p2.calcPossiblePerms(6) == 23040


def test_config_parser(tmp_path):
    from hic_genome_assembler_amd import run_hicAssembler as run, synth
    paths = {k: "/data/" + k for k in ("hicProBedFile", "hicProBiasFile", "hicProMatrixFile", "hicProScaffSizeFile")}
    cfg = synth.write_config(str(tmp_path / "c.txt"), paths, str(tmp_path / "o"), str(tmp_path / "p"), 100000,
                             modularity=0.0, psig=0.01)
    v = run.readConfigFileToVariables(cfg)
    assert v["resolution"] == 100000 and v["psig"] == 0.01 and v["modularity"] == 0.0
    assert v["binGroupFile"] == str(tmp_path / "o") + "/binGroups.txt"          # prefixed at parse time (RUN:95-96)
    assert v["avgClusterPlot"] == str(tmp_path / "p") + "/avgCluster.png"
    assert v["hyperGeom"] is True and v["hmm"] is False and v["nScaffolds"] == 6
    assert run.ensureAllVariablesAreSet(v) is False
    # an empty value leaves the key unset -> refuse to run, also for part3/4 keys (RUN:221-239)
    text = open(cfg).read().replace("validPairFile = /dev/null", "validPairFile = ")
    (tmp_path / "c2.txt").write_text(text)
    assert run.ensureAllVariablesAreSet(run.readConfigFileToVariables(str(tmp_path / "c2.txt"))) is True
    # both strategies True -> refuse (RUN:230,241); bad numbers keep defaults; values keep leading blanks
    text = open(cfg).read().replace("hmm = False", "hmm = true").replace("minSize = 5", "minSize = five") \
        .replace("chromosomePlotSuffix = synthetic", "chromosomePlotSuffix =  500 Kb")
    (tmp_path / "c3.txt").write_text(text + "\nthis line has no separator\n")
    v3 = run.readConfigFileToVariables(str(tmp_path / "c3.txt"))
    assert v3["hmm"] is True and v3["minSize"] == 5 and v3["chromosomePlotSuffix"] == " 500 Kb"
    assert run.ensureAllVariablesAreSet(v3) is True
    # documented difference: hmm = True with hyperGeom = False (the reference's hmmlearn path, out of scope) is refused
    # HERE, before any part runs - and a config line without ' = ' is skipped with a warning where the reference's
    # parser raises IndexError: c3.txt above ended with such a line and still parsed
    text = open(cfg).read().replace("hmm = False", "hmm = True").replace("hyperGeom = True", "hyperGeom = False")
    (tmp_path / "c4.txt").write_text(text)
    v4 = run.readConfigFileToVariables(str(tmp_path / "c4.txt"))
    assert v4["hmm"] is True and v4["hyperGeom"] is False
    assert run.ensureAllVariablesAreSet(v4) is True
    with pytest.raises(SystemExit) as ex:
        run.main(["-part1", "-c", str(tmp_path / "c4.txt")])
    assert ex.value.code in (None, 0)                             # the reference's silent sys.exit() (RUN:270-271)
    assert not os.path.exists(str(tmp_path / "o" / "dendrogramOrder.txt"))


def test_loaders_match_oracle(tmp_path):
    from hic_genome_assembler_amd import hostio
    paths = gc.write_case_files("n300_edges", str(tmp_path))
    bins = hostio.initiateLoci(paths["hicProBedFile"], paths["hicProBiasFile"])
    bins_o = orc.initiate_loci(paths["hicProBedFile"], paths["hicProBiasFile"])
    assert [(b.ID, b.chrom, b.start, b.stop, b.bias) for b in bins] == \
        [(b.ID, b.chrom, b.start, b.stop, b.bias) for b in bins_o]
    assert len(bins) == 298                                      # two "nan" bias lines dropped (S2C:57)
    ref = orc.build_adjacency(paths["hicProMatrixFile"], bins_o)
    for engine in ("native", "pandas"):
        m = hostio.read_contact_matrix(paths["hicProMatrixFile"], bins, engine=engine)
        assert np.array_equal(m, ref), engine
    # duplicates: the later line wins, unknown IDs are skipped (S2C:84-89)
    dup = tmp_path / "dup.matrix"
    dup.write_text("1\t2\t5.0\n2\t1\t7.5\n1\t999999\t3.0\n3\t3\t1.25\n1\t2\t9.0\n2\t3\t0.1\n")
    some = bins[:3]
    for engine in ("native", "pandas"):
        m2 = hostio.read_contact_matrix(str(dup), some, engine=engine)
        assert np.array_equal(m2, orc.build_adjacency(str(dup), some)), engine
    # what Python's split/int/float accept: CRLF, exponents, a fourth column, blanks around the value, no final newline
    odd = tmp_path / "odd.matrix"
    odd.write_text("1\t2\t5.0\r\n2\t3\t1e-3\n3\t1\t.5\textra\n2\t2\t 4.25 \n1\t1\t+7\n3\t3\t1E2")
    assert np.array_equal(hostio.read_contact_matrix(str(odd), some), orc.build_adjacency(str(odd), some))
    # many duplicates spread over the file: every thread boundary must still honour "the later line wins"
    rng = np.random.default_rng(0)
    big = tmp_path / "big.matrix"
    ids = [b.ID for b in bins[:40]]
    with open(big, "w") as fh:
        for _ in range(60000):
            a, b = rng.choice(ids, 2)
            fh.write("%d\t%d\t%r\n" % (a, b, float(rng.random())))
    forty = bins[:40]
    assert np.array_equal(hostio.read_contact_matrix(str(big), forty), orc.build_adjacency(str(big), forty))
    # number formats: HiC-Pro's 6 decimals (exact fast path), short/long significands, exponents, tiny and huge values
    fmt = tmp_path / "formats.matrix"
    with open(fmt, "w") as fh:
        k = 0
        for a in ids:
            for b in ids:
                if b < a:
                    continue
                x = float(rng.random() * 10 ** int(rng.integers(-12, 12)))
                text = ["%.6f" % x, "%.3e" % x, "%r" % x, "%.15g" % x, "%.17g" % x, "%d" % int(x), "%.20f" % x,
                        "%.1f" % x, "%e" % (x * 1e-300), "%e" % (x * 1e290)][k % 10]
                fh.write("%d\t%d\t%s\n" % (a, b, text))
                k += 1
    assert np.array_equal(hostio.read_contact_matrix(str(fmt), forty), orc.build_adjacency(str(fmt), forty))
    bad = tmp_path / "bad.matrix"
    bad.write_text("1\t2\t5.0\n\n2\t3\t1.0\n")
    from hic_genome_assembler_amd import _lib
    with pytest.raises(_lib.HicmiError):
        hostio.read_contact_matrix(str(bad), some)
    subset = {b.ID: '' for b in bins[10:20]}
    sub = hostio.initiateLoci(paths["hicProBedFile"], paths["hicProBiasFile"], binID_dict=subset)
    assert [b.ID for b in sub] == [b.ID for b in bins[10:20]]


def test_part1_rejects_unimplemented_strategies(tmp_path):
    from hic_genome_assembler_amd import scaffoldToChromosomes as p1
    args = ["x"] * 10
    with pytest.raises(NotImplementedError):                       # hmm = True (hmmlearn's stochastic EM)
        p1.runPipeline(*args, False, True, 5, 0.0, 20, .05, 5, .2, 100000)


def test_intermediate_files_round_trip(tmp_path):
    """runResident keeps the in-memory values instead of parsing its own intermediate files back (the
    reference re-reads them, S2C:1124/1147): what the readers return for those files must be the same."""
    from hic_genome_assembler_amd import scaffoldToChromosomes as s2c
    from hic_genome_assembler_amd.hostio import Bin
    rng = np.random.default_rng(5)
    bins = [Bin(10 + i, "scaf%d" % (i // 7), 1000 * i, 1000 * (i + 1), float(rng.random()), 0.0) for i in range(60)]
    dend = {"ivl": [b.chrom + "_" + str(b.ID) for b in bins], "leaves": rng.permutation(60).tolist()}
    s2c.dendrogramLeafOrder_toFile(dend, str(tmp_path / "d.txt"))
    back = s2c.readDengrogramLeavesFromFile(str(tmp_path / "d.txt"))
    assert back["leaves"] == dend["leaves"] and back["ivl"] == dend["ivl"]
    for cuts in ([], [9], [9, 30, 31]):
        groups = s2c.writeBinGroupingsToFile(cuts, bins, str(tmp_path / "g.txt"))
        assert s2c.readBinGroupingsFromFile(str(tmp_path / "g.txt")) == groups
        assert [s2c._pairs_of_lines(g) for g in groups] == s2c._bin_group_pairs(cuts, bins)
        assert sum(len(g) for g in groups) == 60 and len(groups) == len(cuts) + 1
    # the voting report: per-group helper and whole-genome function agree with a direct restatement
    groups = s2c.writeBinGroupingsToFile([9, 30], bins, str(tmp_path / "g.txt"))
    final = s2c.assessChromosomeClustering(groups, str(tmp_path / "a.txt"))
    sizes = {"scaf%d" % k: 7000 for k in range(9)}
    s2c.writeChromosomeGroupingsToFile(final, sizes, str(tmp_path / "c.txt"))
    kept = [int(l.split("\t")[0]) for l in open(tmp_path / "c.txt") if not l.startswith("#")]
    # scaf1 has 2 of 7 bins in group 1 and 5 of 7 (71 %) in group 2; scaf4 2/7 vs 5/7: each joins one group once
    assert sorted(kept) == [b.ID for b in bins]
    report = open(tmp_path / "a.txt").read()
    assert "scaf1\t2\t7\t28.57%" in report and "scaf1\t5\t7\t71.43%" in report
    assert "Falsely clustered nodes 4" in report
    # the pre-formatted forms runResident hands to the writer thread give the same bytes as formatting in the writers
    labels = [b.chrom + "_" + str(b.ID) for b in bins]
    s2c.dendrogramLeafOrder_toFile({"ivl": [labels[i] for i in dend["leaves"]], "leaves": dend["leaves"]}, str(tmp_path / "d3.txt"))
    s2c.dendrogramLeafOrder_toFile({"ivl": None, "leaves": dend["leaves"]}, str(tmp_path / "d4.txt"),
                                   [lab + "\t" + str(i) for i, lab in enumerate(labels)])
    assert open(tmp_path / "d3.txt").read() == open(tmp_path / "d4.txt").read()
    s2c.writeBinGroupingsToFile([9, 30], bins, str(tmp_path / "g2.txt"), {b.ID: s2c._bin_line(b) for b in bins})
    assert open(tmp_path / "g.txt").read() == open(tmp_path / "g2.txt").read()
    s2c.writeChromosomeGroupingsToFile(final, sizes, str(tmp_path / "c2.txt"),
                                       {int(b.ID): str(int(b.ID)) + "\t" + str(b.chrom) + "\n" for b in bins})
    assert open(tmp_path / "c.txt").read() == open(tmp_path / "c2.txt").read()


def test_part2_file_text_by_pieces(tmp_path):
    """orderGenome formats each chromosome's share of the two output files on its scan thread; written from those pieces
    the files are byte for byte what the writers produce from the scaffold lists."""
    from hic_genome_assembler_amd import orderGenome as og
    import contextlib, io
    groups, k = [], 0
    for c in range(3):
        g = []
        for i in range(4 + c):
            sc = og.Scaffold("s%d_%d" % (c, i), list(range(k, k + 1 + (i % 3))), "-" if (i + c) % 2 else "+")
            k += len(sc.binList)
            g.append(sc)
        groups.append(g)
    groups.append([og.Scaffold("empty", [], "+")])
    flat = [s for g in groups for s in g]
    with contextlib.redirect_stdout(io.StringIO()) as out_a:
        og.writeScaffoldOrderingsToFile(groups, str(tmp_path / "o1.txt"))
        og.writeBinIDsOrderingToFile(flat, str(tmp_path / "p1.txt"))
    with contextlib.redirect_stdout(io.StringIO()) as out_b:
        og.writeScaffoldOrderingsToFile(groups, str(tmp_path / "o2.txt"), [og._scaffold_lines(g) for g in groups])
        og.writeBinIDsOrderingToFile(flat, str(tmp_path / "p2.txt"), [og._bin_rows(g) for g in groups])
    assert open(tmp_path / "o1.txt").read() == open(tmp_path / "o2.txt").read()
    assert open(tmp_path / "p1.txt").read() == open(tmp_path / "p2.txt").read()
    assert out_a.getvalue() == out_b.getvalue()
    text = open(tmp_path / "p1.txt").read()
    assert text.startswith("#ScaffoldID\tHiCPro-BinID\ns0_0\t0") and not text.endswith("\n")


def test_matrix_binary_cache(tmp_path):
    """the parsed matrix is cached as .npy under a key of the text file's size/mtime and the
    bin IDs; a matching key skips the parse, a changed text file or bin list invalidates it."""
    from hic_genome_assembler_amd import hostio, synth
    lay = synth.make_layout(120, seed=3)
    c = synth.dense_contacts(lay, seed=3, sinkhorn_iters=4)
    files = synth.write_hicpro(str(tmp_path), lay, c, "m")
    paths = {"matrix": files["hicProMatrixFile"]}
    bins = hostio.initiateLoci(files["hicProBedFile"], files["hicProBiasFile"])
    cache_dir = tmp_path / "cache"
    first = np.array(hostio.read_contact_matrix_cached(paths["matrix"], bins, str(cache_dir)))
    assert np.array_equal(first, hostio.read_contact_matrix(paths["matrix"], bins))
    npy = cache_dir / (os.path.basename(paths["matrix"]) + ".hicmi.npy")
    assert npy.exists()
    # served from the cache: poison the cached array and see the poison come back
    poisoned = first.copy(); poisoned[0, 0] = -123.0
    np.save(str(npy), poisoned)
    again = hostio.read_contact_matrix_cached(paths["matrix"], bins, str(cache_dir))
    assert again[0, 0] == -123.0 and not again.flags.writeable
    # a different bin list (one bin dropped) must not be served from it
    fewer = hostio.read_contact_matrix_cached(paths["matrix"], bins[1:], str(cache_dir))
    assert fewer.shape == (len(bins) - 1, len(bins) - 1) and np.array_equal(np.asarray(fewer), first[1:, 1:])
    # ... and neither must a modified text file
    with open(paths["matrix"], "a") as fh:
        fh.write("%d\t%d\t7.5\n" % (bins[1].ID, bins[2].ID))
    fresh = hostio.read_contact_matrix_cached(paths["matrix"], bins[1:], str(cache_dir))
    assert fresh[0, 1] == 7.5 and fresh[1, 0] == 7.5


def test_read_chroms_line_handling(tmp_path):
    """OG:216-237 line by line (strip "\\r" then "\\n", '#' lines open a group) against the bulk reader."""
    import contextlib
    import io
    from hic_genome_assembler_amd import orderGenome as og

    def literal(path):
        chroms, cur = [], []
        with open(path) as fh:
            fh.readline()
            for line in fh:
                line = line.strip("\r").strip("\n")
                if line[0] != "#":
                    cols = line.split("\t")
                    cur.append([int(cols[0]), cols[1]])
                else:
                    chroms.append(cur)
                    cur = []
        chroms.append(cur)
        return chroms
    texts = ["### g1 ###\n1\ta\tx\n2\tb\ty\n### g2 ###\n3\tc\tz\n", "### g1 ###\n1\ta\tx\n2\tb",
             "### g1 ###\r\n1\ta\tx\r\n### g2 ###\r\n3\tc\r\n", "### g1 ###\n"]
    for k, text in enumerate(texts):
        path = tmp_path / ("g%d.txt" % k)
        with open(path, "w", newline="") as fh:
            fh.write(text)
        with contextlib.redirect_stdout(io.StringIO()):
            assert og.readChromsFromFile(str(path)) == literal(str(path)), text
