"""CPU oracle for Parts 3 and 4 (orientSmallScaffolds.py = OSS, writeAssembledFasta.py = WAF).

TEST INFRASTRUCTURE - only tests/ may import this.  Plain-loop restatement of the two reference modules, line
references in the docstrings; pinned by tests/golden/part34/ (outputs the reference itself wrote, see
oracle/gen_golden_part34.py): tests/test_part34_cpu.py checks this file against them, then uses it as the
checker for randomised cases the golden files do not cover.
"""
import math


class Scaf:
    def __init__(self, name, orientation):
        self.name, self.orientation, self.size, self.sites, self.bins = name, orientation, 0.0, [], None

    def site_counts(self, cutoff):
        """OSS:17-31: (left, right); a site is 'left' when <= cutoff, otherwise 'right' when > size - cutoff."""
        left = right = 0
        for c in self.sites:
            if c <= cutoff:
                left += 1
            elif c > self.size - cutoff:
                right += 1
        return left or 1, right or 1


def load(order_file, size_file, site_file, resolution):
    """OSS:34-107."""
    groups, cur, by_name = [], [], {}
    with open(order_file) as fh:
        fh.readline()
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] == '#':
                groups.append(cur)
                cur = []
            else:
                cols = line.split('\t')
                by_name[cols[0]] = Scaf(cols[0], cols[1])
                cur.append(by_name[cols[0]])
    groups.append(cur)
    with open(size_file) as fh:
        for line in fh:
            cols = line.strip('\r').strip('\n').split('\t')
            if cols[0] in by_name:
                by_name[cols[0]].size = float(cols[1])
                by_name[cols[0]].bins = math.ceil(float(cols[1]) / float(resolution))
    with open(site_file) as fh:
        for line in fh:
            cols = line.strip('\r').strip('\n').split('\t')
            if cols[0] in by_name:
                by_name[cols[0]].sites.append(int(cols[2]))
    for s in by_name.values():
        s.sites.sort()
    return groups, by_name


def triplets_of(group):
    """OSS:109-137."""
    out = []
    for i, s in enumerate(group):
        if s.bins != 1:
            continue
        left = group[i - 1] if i != 0 else None
        right = group[i + 1] if i <= len(group) - 2 else None
        if left is not None and right is not None:
            out.append([left, s, right])
        elif left is None and right is not None:
            out.append([s, right])
        elif left is not None and right is None:
            out.append([left, s])
    return out


def scan_pairs(pair_file, keys):
    """OSS:159-177: {(s1, s2): [(pos1, pos2), ...]} for the registered ordered pairs."""
    found = {k: [] for k in keys}
    with open(pair_file) as fh:
        for line in fh:
            cols = line.strip('\r').strip('\n').split('\t')
            k = (cols[1], cols[4])
            if k in found:
                found[k].append((int(cols[2]), int(cols[5])))
    return found


def _links(found, a, b):
    """(list, index of a's position in a row) - (a, b) first, (b, a) only if that is empty (OSS:192-199)."""
    if len(found[(a.name, b.name)]) != 0:
        return found[(a.name, b.name)], 0
    if len(found[(b.name, a.name)]) != 0:
        return found[(b.name, a.name)], 1
    return None, 0


def orient_middle(t, found, cutoff):
    """OSS:179-240."""
    s0, s1, s2 = t
    res = {s.name: s.site_counts(cutoff) for s in t}
    p = m = 0
    rows, ia = _links(found, s1, s2)
    if rows is not None:
        for r in rows:
            c2 = r[1 - ia]
            if s2.orientation == "+":
                p += c2 <= cutoff
            else:
                p += (s2.size - c2) <= cutoff
        p = float(p) / float(res[s1.name][1] + (res[s2.name][0] if s2.orientation == "+" else res[s2.name][1]))
    rows, ia = _links(found, s1, s0)
    if rows is not None:
        for r in rows:
            c0 = r[1 - ia]
            if s0.orientation == "-":
                m += c0 <= cutoff
            else:
                m += (s0.size - c0) <= cutoff
        m = float(m) / float(res[s1.name][1] + (res[s0.name][0] if s0.orientation == "-" else res[s0.name][1]))
    return s1.name, "+" if p >= m else "-"


def orient_left_edge(left, right, found, cutoff):
    """OSS:242-288."""
    l_res, r_res = left.site_counts(float(left.size / 2.)), right.site_counts(cutoff)
    p = m = 0
    rows, il = _links(found, left, right)
    if rows is not None:
        lo, hi = (0, cutoff) if right.orientation == "+" else (right.size - cutoff, right.size)
        for r in rows:
            cl, cr = r[il], r[1 - il]
            if cl >= float(left.size / 2.) and lo <= cr <= hi:
                p += 1
            elif lo <= cr <= hi:
                m += 1
    rs = r_res[0] if right.orientation == "+" else r_res[1]
    p, m = float(p) / float(l_res[1] + rs), float(m) / float(l_res[0] + rs)
    return left.name, "+" if p >= m else "-"


def orient_right_edge(left, right, found, cutoff):
    """OSS:290-336."""
    l_res, r_res = left.site_counts(cutoff), right.site_counts(float(right.size / 2.))
    p = m = 0
    rows, il = _links(found, left, right)
    if rows is not None:
        lo, hi = (left.size - cutoff, left.size) if left.orientation == "+" else (0, cutoff)
        for r in rows:
            cl, cr = r[il], r[1 - il]
            if cr < float(right.size / 2.) and lo <= cl <= hi:
                p += 1
            elif lo <= cl <= hi:
                m += 1
    ls = l_res[1] if left.orientation == "+" else l_res[0]
    p, m = float(p) / float(ls + r_res[0]), float(m) / float(ls + r_res[1])
    return right.name, "+" if p >= m else "-"


def run_part3(order_file, size_file, site_file, pair_file, out_file, cutoff, resolution):
    """OSS:370-433."""
    groups, by_name = load(order_file, size_file, site_file, resolution)
    trips = [triplets_of(g) for g in groups]
    keys = set()
    for per_chrom in trips:
        for t in per_chrom:
            for a, b in zip(t, t[1:]):
                keys.add((a.name, b.name))
                keys.add((b.name, a.name))
    found = scan_pairs(pair_file, keys)
    if cutoff < resolution:
        cutoff = resolution
    with open(out_file, "w") as fh:
        for n, (per_chrom, group) in enumerate(zip(trips, groups), 1):
            for t in per_chrom:
                if len(t) == 3:
                    name, o = orient_middle(t, found, cutoff)
                elif t[0].name == group[0].name:
                    name, o = orient_left_edge(t[0], t[1], found, cutoff)
                else:
                    name, o = orient_right_edge(t[0], t[1], found, cutoff)
                by_name[name].orientation = o
            fh.write("### Chromosome grouping " + str(n) + " ###\n")
            for s in group:
                fh.write(s.name + "\t" + s.orientation + "\n")


# ---- Part 4 ---------------------------------------------------------------------------------------------
_OPP = {"A": "T", "T": "A", "a": "t", "t": "a", "G": "C", "C": "G", "g": "c", "c": "g", "N": "N", "n": "n"}


def run_part4(fasta_file, order_file, out_file, per_line=50, gap=100):
    """WAF:10-127 (plain-text FASTA only)."""
    seqs, name = {}, None
    with open(fasta_file) as fh:
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] == '>':
                name = line[1:]
                seqs[name] = []
            else:
                seqs[name].append(line)
    seqs = {k: ''.join(v) for k, v in seqs.items()}
    groups, cur = [], []
    with open(order_file) as fh:
        fh.readline()
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] != '#':
                cur.append(line.split('\t')[:2])
            else:
                groups.append(cur)
                cur = []
    groups.append(cur)
    done = set()
    with open(out_file, "w") as out:
        def dump(seq):
            for i in range(0, len(seq), per_line):
                out.write(seq[i:i + per_line] + "\n")
        for n, g in enumerate(groups, 1):
            out.write(">Chr_" + str(n) + "\n")
            joined = []
            for k, (name, o) in enumerate(g):
                done.add(name)
                joined.append(seqs[name] if o == "+" else ''.join(_OPP[c] for c in reversed(seqs[name])))
                if k != len(g) - 1:
                    joined.append("N" * gap)
            dump(''.join(joined))
        for name, seq in seqs.items():
            if name not in done:
                out.write(">" + name + "\n")
                dump(seq)
