"""Parts 3 and 4: host-side text processing - orientSmallScaffolds.py and writeAssembledFasta.py of the
package against (1) the files the REFERENCE wrote for three cases (tests/golden/part34/, made by
oracle/gen_golden_part34.py), (2) the plain-loop oracle on randomised cases, and the native valid-pair scanner's own
edge cases.  No GPU involved: libhicmi's scanner is host code."""
import gzip
import hashlib
import io
import json
import os
import sys
import contextlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import gen_golden_part34 as gen          # noqa: E402  (inputs_for(): the seeded input generator of the golden cases)
import hic_oracle_part34 as orc34        # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "part34")


def _golden(case):
    with open(os.path.join(GOLD, case, "meta.json")) as fh:
        meta = json.load(fh)
    with open(os.path.join(GOLD, case, "finalOrderings.txt")) as fh:
        return meta, fh.read()


@pytest.mark.parametrize("case", sorted(gen.CASES))
def test_parts34_match_the_reference_outputs(case, tmp_path):
    from hic_genome_assembler_amd import orientSmallScaffolds as p3, writeAssembledFasta as p4
    meta, final_text = _golden(case)
    inp = gen.inputs_for(case, str(tmp_path))
    for k, digest in meta["inputs_sha256"].items():          # the generator still makes the files the goldens were made from
        assert hashlib.sha256(open(inp[k], "rb").read()).hexdigest() == digest, k
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        p3.runPipeline(inp["order"], inp["sizes"], inp["restrictionSiteFile"], inp["validPairFile"],
                       str(tmp_path / "final.txt"), inp["cutoff"], inp["resolution"])
        p4.runPipeline(inp["originalFastaFile"], str(tmp_path / "final.txt"), str(tmp_path / "assembled.fasta"))
    assert open(tmp_path / "final.txt").read() == final_text
    data = open(tmp_path / "assembled.fasta", "rb").read()
    assert len(data) == meta["assembled_bytes"] and hashlib.sha256(data).hexdigest() == meta["assembled_sha256"]
    stats = [l for l in log.getvalue().splitlines() if l.startswith("Total ") and "run-time" not in l]
    assert stats == meta["stats"]
    # the oracle is pinned by the same files
    orc34.run_part3(inp["order"], inp["sizes"], inp["restrictionSiteFile"], inp["validPairFile"], str(tmp_path / "o3.txt"),
                    inp["cutoff"], inp["resolution"])
    assert open(tmp_path / "o3.txt").read() == final_text
    orc34.run_part4(inp["originalFastaFile"], str(tmp_path / "o3.txt"), str(tmp_path / "o4.fasta"))
    assert open(tmp_path / "o4.fasta", "rb").read() == data


def _random_case(rng, work):
    """A genome of a few chromosomes in which most scaffolds are one bin long, links that are sparse enough for
    empty pair lists and ties, both mate orders, windows that straddle the cut-off."""
    res = 10000
    names, sizes, groups = [], {}, []
    for c in range(int(rng.integers(1, 5))):
        grp = []
        for _ in range(int(rng.integers(1, 7))):
            name = "s%d" % len(names)
            names.append(name)
            sizes[name] = int(rng.integers(2000, 3 * res)) if rng.random() < 0.7 else int(rng.integers(res + 1, 8 * res))
            grp.append((name, "+" if rng.random() < 0.5 else "-"))
        groups.append(grp)
    os.makedirs(work, exist_ok=True)
    p = {k: os.path.join(work, k) for k in ("order", "sizes", "sites", "pairs")}
    with open(p["order"], "w") as fh:
        for c, grp in enumerate(groups, 1):
            fh.write("### Chromosome grouping %d ###\n" % c)
            fh.write("".join("%s\t%s\n" % g for g in grp))
    with open(p["sizes"], "w") as fh:
        fh.write("".join("%s\t%d\n" % (n, sizes[n]) for n in names) + "other\t5000\n")
    with open(p["sites"], "w") as fh:
        for n in names + ["other"]:
            for _ in range(int(rng.integers(0, 12))):
                fh.write("%s\t0\t%d\tx\t0\t+\n" % (n, int(rng.integers(1, sizes.get(n, 5000) + 1))))
    with open(p["pairs"], "w") as fh:
        flat = [g[0] for grp in groups for g in grp] + ["other"]
        for i in range(int(rng.integers(0, 400))):
            a = flat[int(rng.integers(len(flat)))]
            b = flat[min(len(flat) - 1, max(0, flat.index(a) + int(rng.integers(-1, 2))))]
            fh.write("r%d\t%s\t%d\t+\t%s\t%d\t-\t1\tx\ty\t1\t1\n"
                     % (i, a, int(rng.integers(1, sizes.get(a, 5000) + 1)), b, int(rng.integers(1, sizes.get(b, 5000) + 1))))
    return p, res


def test_part3_matches_the_oracle_on_random_genomes(tmp_path):
    from hic_genome_assembler_amd import orientSmallScaffolds as p3
    rng = np.random.default_rng(17)
    for trial in range(40):
        p, res = _random_case(rng, str(tmp_path / ("t%d" % trial)))
        cutoff = int(rng.choice([res // 2, res, 3 * res]))          # below the resolution it is raised to it (OSS:378-380)
        with contextlib.redirect_stdout(io.StringIO()):
            p3.runPipeline(p["order"], p["sizes"], p["sites"], p["pairs"], p["order"] + ".gpu", cutoff, res)
        orc34.run_part3(p["order"], p["sizes"], p["sites"], p["pairs"], p["order"] + ".orc", cutoff, res)
        assert open(p["order"] + ".gpu").read() == open(p["order"] + ".orc").read(), trial


def test_valid_pair_scanner(tmp_path):
    from hic_genome_assembler_amd import _lib
    names = ["a", "bb", "c c"]
    pairs = [(0, 1), (1, 0), (2, 2)]
    text = ("r\ta\t10\t+\tbb\t20\t-\tmore\tcolumns\n"
            "r\tbb\t 7 \t+\ta\t8\n"                               # int() tolerates blanks; six columns are enough
            "r\ta\tnot-a-number\t+\tzz\t1\t-\n"                   # unregistered pair: its positions are never parsed
            "r\tc c\t5\t+\tc c\t6\t-\r\n"
            "r\ta\t1\t+\ta\t2\t-\n")                              # (a, a) is not registered
    f = tmp_path / "p.txt"
    f.write_text(text)
    for threads in (1, 4):
        idx, p1, p2, n_lines = _lib.scan_valid_pairs(str(f), names, pairs, threads)
        assert n_lines == 5 and idx.tolist() == [0, 1, 2] and p1.tolist() == [10, 7, 5] and p2.tolist() == [20, 8, 6]
    big = tmp_path / "big.txt"
    rng = np.random.default_rng(1)
    rows = [(names[int(a)], int(x), names[int(b)], int(y)) for a, b, x, y in rng.integers(0, 3, size=(20000, 4)) * [1, 1, 50, 50]]
    big.write_text("".join("r\t%s\t%d\t+\t%s\t%d\t-\n" % r for r in rows))
    want = [(pairs.index((names.index(a), names.index(b))), x, y) for a, x, b, y in rows if (names.index(a), names.index(b)) in pairs]
    for threads in (1, 0):
        idx, p1, p2, n_lines = _lib.scan_valid_pairs(str(big), names, pairs, threads)
        assert n_lines == len(rows) and list(zip(idx.tolist(), p1.tolist(), p2.tolist())) == want       # file order
    bad = tmp_path / "bad.txt"
    bad.write_text("r\ta\t10\t+\tbb\n")                         # five columns and a registered pair: cols[5] -> IndexError
    with pytest.raises(_lib.HicmiError):
        _lib.scan_valid_pairs(str(bad), names, pairs)
    bad.write_text("r\ta\t10\t+\n")                              # four columns: cols[4] -> IndexError on any line
    with pytest.raises(_lib.HicmiError):
        _lib.scan_valid_pairs(str(bad), names, pairs)
    short = tmp_path / "short.txt"                                # five columns, no registered pair: the reference never
    short.write_text("r\ta\t10\t+\tzz\nr\ta\t3\t+\tbb\t4\n")     # touches cols[5] there (orientSmallScaffolds.py:168-170)
    idx, p1, p2, n_lines = _lib.scan_valid_pairs(str(short), names, pairs)
    assert n_lines == 2 and idx.tolist() == [0] and p1.tolist() == [3] and p2.tolist() == [4]
    bad.write_text("r\ta\t1x\t+\tbb\t3\t-\n")
    with pytest.raises(_lib.HicmiError):
        _lib.scan_valid_pairs(str(bad), names, pairs)
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    assert _lib.scan_valid_pairs(str(empty), names, pairs)[3] == 0


def test_part4_details(tmp_path):
    from hic_genome_assembler_amd import writeAssembledFasta as p4
    assert p4.reverseTranscribeSeq("AaCcGgTtNn") == "nNaAcCgGtT"
    with pytest.raises(KeyError) as err:
        p4.reverseTranscribeSeq("ACRGTY")                          # the reversed walk meets Y first
    assert err.value.args[0] == "Y"
    fasta = tmp_path / "g.fasta.gz"
    with gzip.open(fasta, "wt") as fh:
        fh.write(">s1\n" + "A" * 60 + "\n" + "C" * 40 + "\n>s2\nGGT\n>empty\n>s3\nTTTT\n")
    seqs = p4.readFastaIntoMem(str(fasta))
    assert seqs == {"s1": "A" * 60 + "C" * 40, "s2": "GGT", "empty": "", "s3": "TTTT"}
    order = tmp_path / "o.txt"
    order.write_text("### Chromosome grouping 1 ###\ns2\t-\ns1\t+\n### Chromosome grouping 2 ###\ns3\t+\n")
    with contextlib.redirect_stdout(io.StringIO()) as log:
        p4.writeNewFasta(p4.readChromosomeOrderingFile(str(order)), seqs, str(tmp_path / "out.fa"), charsPerLine=50, nGapLength=100)
    joined = "ACC" + "N" * 100 + "A" * 60 + "C" * 40           # 203 bases: 4 full lines of 50 and one of 3
    want = ">Chr_1\n" + "".join(joined[i:i + 50] + "\n" for i in range(0, 203, 50)) + ">Chr_2\nTTTT\n>empty\n"
    assert open(tmp_path / "out.fa").read() == want
    assert "Total new gaps introduced\t1" in log.getvalue() and "Total ungrouped scaffolds\t1" in log.getvalue()
    exact = tmp_path / "e.fa"
    with open(exact, "w") as fh:
        p4.writeSeqToFile(fh, "G" * 100, charsPerLine=50)          # a multiple of the line length: no empty last line
    assert open(exact).read() == "G" * 50 + "\n" + "G" * 50 + "\n"


def test_cli_runs_parts_3_and_4(tmp_path):
    from hic_genome_assembler_amd import run_hicAssembler, synth
    case = "n160"
    meta, final_text = _golden(case)
    inp = gen.inputs_for(case, str(tmp_path / "in"))
    import golden_cases as gc
    paths = gc.write_case_files(case, str(tmp_path / "in"))
    cfg = synth.write_config(str(tmp_path / "config.txt"), paths, str(tmp_path / "out"), str(tmp_path / "plots"), 100000)
    text = open(cfg).read()
    for key in ("restrictionSiteFile", "validPairFile", "originalFastaFile"):
        text = text.replace("%s = /dev/null" % key, "%s = %s" % (key, inp[key]))
    open(cfg, "w").write(text)
    with open(tmp_path / "out" / "chromosomeOrders.txt", "w") as fh:
        fh.write(gc.golden_text(case, "chromosomeOrders.txt"))
    with contextlib.redirect_stdout(io.StringIO()):
        run_hicAssembler.main(["-part3", "-part4", "-c", cfg])
    assert open(tmp_path / "out" / "finalOrderings.txt").read() == final_text
    data = open(tmp_path / "out" / "assembled.fasta", "rb").read()
    assert hashlib.sha256(data).hexdigest() == meta["assembled_sha256"]
