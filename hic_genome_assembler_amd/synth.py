"""Synthetic HiC-Pro datasets.

The reference ships no example data (its example config points at the
author's home directory, hicAssembler_config_workingExample.txt:32-41), so
every test, fixture and benchmark in this repository runs on maps made here:

* K chromosomes with linearly decreasing sizes, K = max(8, N // 1400);
* scaffolds of ~Geometric(mean 13) bins, random true orientation, scaffold
  order shuffled in the ``.bed`` file;
* contacts ``1 / (1 + |p_i - p_j|)`` inside a chromosome and ``4e-4`` between
  chromosomes, times symmetric log-normal noise (sigma 0.25);
* symmetric Sinkhorn balancing (what ICE converges to) to equal row sums,
  scaled to row sum 1000;
* dense (no zeros) and written with ``repr`` precision so that rows are
  tie-free (the reference's ``numpy.argsort`` tie order is undefined,
  SURVEY.md section 8c).

Two producers share the layout code: :func:`dense_contacts` (NumPy, used for
files and fixtures) and :func:`dense_contacts_torch` (generates directly in
device memory for ``bench.py``; different random stream, same distribution).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np


@dataclass
class Layout:
    """Per-bin description of a synthetic genome, in ``.bed`` order."""
    n_bins: int
    resolution: int
    scaffold_of_bin: np.ndarray    # int32[N] scaffold index (into scaffold_names)
    start: np.ndarray              # int64[N] bp start of the bin inside its scaffold
    stop: np.ndarray               # int64[N]
    chrom_of_bin: np.ndarray       # int32[N] planted chromosome
    pos_of_bin: np.ndarray         # int32[N] planted position along the chromosome
    scaffold_names: list
    scaffold_sizes_bp: np.ndarray  # int64[S]
    scaffold_chrom: np.ndarray     # int32[S] planted chromosome of each scaffold
    scaffold_rank: np.ndarray      # int32[S] planted rank of the scaffold inside its chromosome
    scaffold_orient: np.ndarray    # int8[S]  +1 / -1 planted orientation

    @property
    def bin_ids(self) -> np.ndarray:
        return np.arange(1, self.n_bins + 1, dtype=np.int64)   # HiC-Pro IDs are 1-based


def chromosome_sizes(n_bins: int, n_chrom: int) -> np.ndarray:
    """Linearly decreasing chromosome sizes that sum to n_bins."""
    w = np.arange(2 * n_chrom, n_chrom, -1, dtype=np.float64)
    sizes = np.floor(w / w.sum() * n_bins).astype(np.int64)
    sizes[0] += n_bins - sizes.sum()
    return sizes


def make_layout(n_bins: int, seed: int = 1, n_chrom: int | None = None,
                mean_scaffold_bins: float = 13.0, resolution: int = 100000) -> Layout:
    rng = np.random.default_rng(seed)
    if n_chrom is None:
        n_chrom = max(8, n_bins // 1400)
    n_chrom = max(1, min(n_chrom, n_bins))
    csizes = chromosome_sizes(n_bins, n_chrom)

    scaffolds = []                   # (chrom, rank, orient, n_bins_in_scaffold)
    for c, size in enumerate(csizes):
        left, rank = int(size), 0
        while left > 0:
            ln = int(min(left, rng.geometric(1.0 / mean_scaffold_bins)))
            scaffolds.append((c, rank, 1 if rng.random() < 0.5 else -1, ln))
            left -= ln
            rank += 1
    order = rng.permutation(len(scaffolds))      # order of scaffolds in the .bed file

    scaffold_of_bin = np.empty(n_bins, np.int32)
    start = np.empty(n_bins, np.int64)
    stop = np.empty(n_bins, np.int64)
    chrom_of_bin = np.empty(n_bins, np.int32)
    pos_of_bin = np.empty(n_bins, np.int32)
    # planted offset of each scaffold inside its chromosome
    offset = {}
    acc = {}
    for c, rank, _o, ln in scaffolds:
        offset[(c, rank)] = acc.get(c, 0)
        acc[c] = acc.get(c, 0) + ln
    names, sizes_bp, s_chrom, s_rank, s_orient = [], [], [], [], []
    k = 0
    for new_id, old in enumerate(order):
        c, rank, orient, ln = scaffolds[old]
        names.append("scaffold_%d" % (new_id + 1))
        sizes_bp.append(ln * resolution)
        s_chrom.append(c); s_rank.append(rank); s_orient.append(orient)
        base = offset[(c, rank)]
        for b in range(ln):
            scaffold_of_bin[k] = new_id
            start[k] = b * resolution
            stop[k] = (b + 1) * resolution
            chrom_of_bin[k] = c
            pos_of_bin[k] = base + (b if orient > 0 else ln - 1 - b)
            k += 1
    return Layout(n_bins, resolution, scaffold_of_bin, start, stop, chrom_of_bin, pos_of_bin,
                  names, np.asarray(sizes_bp, np.int64), np.asarray(s_chrom, np.int32),
                  np.asarray(s_rank, np.int32), np.asarray(s_orient, np.int8))


def dense_contacts(layout: Layout, seed: int = 1, sigma: float = 0.25, inter: float = 4e-4,
                   sinkhorn_iters: int = 40, row_sum: float = 1000.0) -> np.ndarray:
    """Dense symmetric fp64 contact map in ``.bed`` order."""
    n = layout.n_bins
    rng = np.random.default_rng(seed + 7919)
    p = layout.pos_of_bin.astype(np.float64)
    same = layout.chrom_of_bin[:, None] == layout.chrom_of_bin[None, :]
    c = np.where(same, 1.0 / (1.0 + np.abs(p[:, None] - p[None, :])), inter)
    noise = rng.normal(0.0, sigma, size=(n, n))
    noise = np.triu(noise) + np.triu(noise, 1).T
    c *= np.exp(noise)
    for _ in range(sinkhorn_iters):
        s = np.sqrt(c.sum(axis=1))
        c /= s[:, None]
        c /= s[None, :]
    c *= row_sum / c.sum(axis=1).mean()
    c = 0.5 * (c + c.T)
    return np.ascontiguousarray(c)


def dense_contacts_torch(layout: Layout, device, seed: int = 1, sigma: float = 0.25,
                         inter: float = 4e-4, sinkhorn_iters: int = 20, row_sum: float = 1000.0):
    """Same distribution as :func:`dense_contacts`, generated in device memory (fp64)."""
    import torch
    n = layout.n_bins
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919)
    p = torch.as_tensor(layout.pos_of_bin, device=device, dtype=torch.float64)
    ch = torch.as_tensor(layout.chrom_of_bin, device=device)
    c = torch.empty((n, n), dtype=torch.float64, device=device)
    blk = max(1, min(n, (1 << 27) // max(n, 1)))          # rows per slab: keep temporaries ~1 GiB
    for r0 in range(0, n, blk):
        r1 = min(n, r0 + blk)
        d = (p[r0:r1, None] - p[None, :]).abs_()
        same = ch[r0:r1, None] == ch[None, :]
        slab = torch.where(same, 1.0 / (1.0 + d), torch.full_like(d, inter))
        z = torch.randn((r1 - r0, n), generator=g, device=device, dtype=torch.float64)
        slab *= torch.exp(z * sigma)
        c[r0:r1] = slab
        del d, same, slab, z
    # symmetrise the noise: keep the upper triangle
    iu = torch.triu_indices(n, n, 1, device=device) if n <= 8192 else None
    if iu is not None:
        c[iu[1], iu[0]] = c[iu[0], iu[1]]
    else:
        for r0 in range(0, n, blk):
            r1 = min(n, r0 + blk)
            cols = torch.arange(n, device=device)[None, :]
            rows = torch.arange(r0, r1, device=device)[:, None]
            lower = cols < rows
            slab = c[r0:r1]
            slab[lower] = c[:, r0:r1].t()[lower]
            del cols, rows, lower
    for _ in range(sinkhorn_iters):
        s = c.sum(dim=1).sqrt_()
        c /= s[:, None]
        c /= s[None, :]
    c *= row_sum / c.sum(dim=1).mean()
    return c


def sparsify(contacts: np.ndarray, zero_fraction: float, decimals: int) -> np.ndarray:
    """What real HiC-Pro maps look like and the dense maps above do not: values rounded to ``decimals`` places (exact
    duplicates everywhere) and the smallest ``zero_fraction`` of the cells set to exactly 0 (similarity exactly 0,
    distance exactly 2.0: long runs of ties in every row).  The diagonal keeps at least 1.0 so no row empties."""
    c = np.round(contacts, decimals)
    c[c <= np.quantile(c, zero_fraction)] = 0.0
    c = 0.5 * (c + c.T)
    np.fill_diagonal(c, np.maximum(np.diag(c), 1.0))
    return np.ascontiguousarray(c)


def write_hicpro(out_dir: str, layout: Layout, contacts, prefix: str = "synth",
                 nan_bias_bins=()) -> dict:
    """Write ``.bed``, ``.biases``, ``.matrix`` (upper-triangle triplets) and the scaffold size file.
    ``contacts=None``: a one-line placeholder ``.matrix`` (for runs that read the matrix from its binary cache,
    hostio.read_contact_matrix_cached - a 16,000-bin text file has 128 M lines).

    Formats follow what the reference's loaders read (scaffoldToChromosomes.py:35-98, 968-979).
    Returns the four paths keyed by the reference's config-variable names.
    """
    os.makedirs(out_dir, exist_ok=True)
    n = layout.n_bins
    ids = layout.bin_ids
    paths = {
        "hicProBedFile": os.path.join(out_dir, prefix + "_abs.bed"),
        "hicProBiasFile": os.path.join(out_dir, prefix + "_iced.matrix.biases"),
        "hicProMatrixFile": os.path.join(out_dir, prefix + "_iced.matrix"),
        "hicProScaffSizeFile": os.path.join(out_dir, prefix + ".sizes"),
    }
    nan_bias = set(int(b) for b in nan_bias_bins)
    with open(paths["hicProBedFile"], "w") as fh:
        for k in range(n):
            fh.write("%s\t%d\t%d\t%d\n" % (layout.scaffold_names[layout.scaffold_of_bin[k]],
                                           layout.start[k], layout.stop[k], ids[k]))
    rng = np.random.default_rng(n)
    bias = 0.5 + rng.random(n)
    with open(paths["hicProBiasFile"], "w") as fh:
        for k in range(n):
            fh.write("nan\n" if int(ids[k]) in nan_bias else repr(float(bias[k])) + "\n")
    with open(paths["hicProMatrixFile"], "w") as fh:
        if contacts is None:
            fh.write("%d\t%d\t1.0\n" % (ids[0], ids[0]))
        for i in range(n if contacts is not None else 0):
            row = contacts[i]
            fh.write("".join("%d\t%d\t%s\n" % (ids[i], ids[j], repr(float(row[j])))
                             for j in range(i, n) if row[j] != 0.0))
    with open(paths["hicProScaffSizeFile"], "w") as fh:
        for name, size in zip(layout.scaffold_names, layout.scaffold_sizes_bp):
            fh.write("%s\t%d\n" % (name, size))
    return paths


def write_config(path: str, hicpro_paths: dict, save_dir: str, plot_dir: str, resolution: int,
                 min_size: int = 5, modularity: float = 0.0, psig: float = 0.05,
                 n_scaffolds: int = 6, scan_scaffolds: int = 5) -> str:
    """Write a config in the reference's ``key = value`` format with every key set
    (run_hicAssembler.py:221-245 refuses empty values even for parts that do not run)."""
    os.makedirs(save_dir, exist_ok=True)
    os.makedirs(plot_dir, exist_ok=True)
    lines = [
        "### synthetic dataset config (same keys as hicAssembler_config.txt) ###",
        "resolution = %d" % resolution,
        "saveFilesDirectory = %s" % save_dir,
        "savePlotsDirectory = %s" % plot_dir,
        "hicProBedFile = %s" % hicpro_paths["hicProBedFile"],
        "hicProBiasFile = %s" % hicpro_paths["hicProBiasFile"],
        "hicProMatrixFile = %s" % hicpro_paths["hicProMatrixFile"],
        "hicProScaffSizeFile = %s" % hicpro_paths["hicProScaffSizeFile"],
        "chromosomeGroupFile = chromosomeGroups.txt",
        "chromosomeOrderFile = chromosomeOrders.txt",
        "finalOrderingsFile = finalOrderings.txt",
        "hyperGeom = True",
        "hmm = False",
        "minSize = %d" % min_size,
        "modularity = %r" % modularity,
        "psig = %r" % psig,
        "convergenceRounds = 5",
        "lookAhead = .2",
        "louvainRounds = 20",
        "dendrogramOrderFile = dendrogramOrder.txt",
        "avgClusterPlot = avgCluster.png",
        "avgClusterPlot_outlined = avgCluster_outlined.png",
        "binGroupFile = binGroups.txt",
        "assessmentFile = assessment.txt",
        "chromosomePlotSuffix = synthetic",
        "fullGenomePlot = fullGenome.png",
        "fullGenomePlotTitle = synthetic genome",
        "plotOrderFile = plotOrder.txt",
        "nScaffolds = %d" % n_scaffolds,
        "scanScaffolds = %d" % scan_scaffolds,
        "lengthCutoff = 500000",
        "restrictionSiteFile = /dev/null",
        "validPairFile = /dev/null",
        "originalFastaFile = /dev/null",
        "assembledFastaFile = assembled.fasta",
    ]
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return path


# ---- inputs of Parts 3 and 4 (orientSmallScaffolds.py, writeAssembledFasta.py) ---------------------------------
def write_part34_inputs(out_dir: str, order_file: str, size_file: str, seed: int = 1, pairs_per_join: int = 400,
                        noise_pairs: int = 5000, site_spacing: int = 4000, fasta_scale: int = 100,
                        prefix: str = "synth") -> dict:
    """Restriction-site, valid-pair and FASTA files for a chromosome-order file written by Part 2.

    * restriction sites (HiC-Pro digest BED: ``scaffold start end name 0 +``): one about every
      ``site_spacing`` bp; the reference reads columns 0 and 2 (orientSmallScaffolds.py:74-83);
    * valid pairs (HiC-Pro allValidPairs: ``read scaf1 pos1 strand1 scaf2 pos2 strand2 ...``): for every pair of
      neighbouring scaffolds ``pairs_per_join`` read pairs drawn near the two facing ends (either mate first),
      plus ``noise_pairs`` between random scaffolds that the scan has to skip; columns 1, 2, 4, 5 are used
      (orientSmallScaffolds.py:159-177);
    * FASTA: random ACGT (some N, some lower case) of ``size / fasta_scale`` bases per scaffold, 60 per line,
      plus two scaffolds that are in no chromosome group.
    Returns the three paths keyed by the reference's config names.
    """
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    sizes = {}
    with open(size_file) as fh:
        for line in fh:
            name, size = line.rstrip("\r\n").split("\t")[:2]
            sizes[name] = int(size)
    groups, cur = [], []
    with open(order_file) as fh:
        fh.readline()
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith("#"):
                groups.append(cur)
                cur = []
            else:
                cur.append(line.split("\t")[:2])
    groups.append(cur)
    paths = {"restrictionSiteFile": os.path.join(out_dir, prefix + "_restriction.bed"),
             "validPairFile": os.path.join(out_dir, prefix + ".allValidPairs"),
             "originalFastaFile": os.path.join(out_dir, prefix + ".fasta")}
    names = list(sizes)
    with open(paths["restrictionSiteFile"], "w") as fh:
        for name in names:
            pos, k = 0, 0
            while True:
                pos += int(rng.integers(site_spacing // 2, site_spacing * 3 // 2))
                if pos >= sizes[name]:
                    break
                fh.write("%s\t%d\t%d\tHIC_%s_%d\t0\t+\n" % (name, max(pos - 4, 0), pos, name, k))
                k += 1
    lines = []
    rid = 0
    for grp in groups:
        for (a, oa), (b, ob) in zip(grp, grp[1:]):
            for _ in range(pairs_per_join):
                # a's end that faces b, b's end that faces a (exponential fall-off into the scaffolds)
                da = min(int(rng.exponential(60000.0)), sizes[a] - 1)
                db = min(int(rng.exponential(60000.0)), sizes[b] - 1)
                pa = sizes[a] - 1 - da if oa == "+" else da
                pb = db if ob == "+" else sizes[b] - 1 - db
                if rng.random() < 0.5:
                    lines.append("read%d\t%s\t%d\t+\t%s\t%d\t-\t300\tf1\tf2\t42\t42" % (rid, a, pa + 1, b, pb + 1))
                else:
                    lines.append("read%d\t%s\t%d\t-\t%s\t%d\t+\t300\tf1\tf2\t42\t42" % (rid, b, pb + 1, a, pa + 1))
                rid += 1
    for _ in range(noise_pairs):
        a, b = names[int(rng.integers(len(names)))], names[int(rng.integers(len(names)))]
        lines.append("read%d\t%s\t%d\t+\t%s\t%d\t-\t300\tf1\tf2\t42\t42"
                     % (rid, a, int(rng.integers(1, sizes[a] + 1)), b, int(rng.integers(1, sizes[b] + 1))))
        rid += 1
    order = rng.permutation(len(lines))
    with open(paths["validPairFile"], "w") as fh:
        fh.write("".join(lines[i] + "\n" for i in order))
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNn", dtype=np.uint8)
    with open(paths["originalFastaFile"], "w") as fh:
        for name in names + ["unplaced_1", "unplaced_2"]:
            n_bases = max(1, sizes.get(name, 1234 * fasta_scale) // fasta_scale)
            seq = alphabet[rng.integers(0, len(alphabet), size=n_bases)].tobytes().decode("ascii")
            fh.write(">" + name + "\n")
            fh.write("".join(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60)))
    return paths
