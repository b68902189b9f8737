"""Command-line driver: ``run_hicAssembler.py [-part1] [-part2] [-part3] [-part4] -config FILE`` - same flags,
same ``key = value`` config format and same file layout as the reference driver
(/root/reference/HIC_ASSEMBLER/run_hicAssembler.py, RUN below), with Parts 1 and 2 executed on
MI355X.  Parts 3 and 4 (read-pair orientation of small scaffolds, FASTA writing) are host-side text
stages (orientSmallScaffolds.py with libhicmi's multi-threaded valid-pair scanner,
writeAssembledFasta.py).

Config rules kept from the reference parser (RUN:9-245):
 * lines are ``name = value`` split on the literal `` = ``; blank lines and lines starting with '#'
   are skipped; values keep any extra leading blanks;
 * file-name keys are prefixed with saveFilesDirectory / savePlotsDirectory AT PARSE TIME, so those
   two keys must come first (RUN:81-98, 181, 185, 215);
 * booleans accept True/true/False/false; malformed numbers keep the default with a warning;
 * every one of the 33 keys must end up non-empty, also those of parts that do not run
   (RUN:221-245); hyperGeom and hmm may not both be True.
One tolerance is added and reported: a non-comment line without `` = `` (both config files shipped
with the reference contain one and crash its parser with IndexError) is skipped with a warning.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

# (key, default, kind, directory key used as prefix or None)
_SPEC = [
    ("resolution", '', "int_required", None),
    ("saveFilesDirectory", '', "str", None),
    ("savePlotsDirectory", '', "str", None),
    ("hicProBedFile", '', "str", None),
    ("hicProBiasFile", '', "str", None),
    ("hicProMatrixFile", '', "str", None),
    ("hicProScaffSizeFile", '', "str", None),
    ("dendrogramOrderFile", '', "str", "saveFilesDirectory"),
    ("avgClusterPlot", '', "str", "savePlotsDirectory"),
    ("avgClusterPlot_outlined", '', "str", "savePlotsDirectory"),
    ("binGroupFile", '', "str", "saveFilesDirectory"),
    ("assessmentFile", '', "str", "saveFilesDirectory"),
    ("hyperGeom", True, "bool", None),
    ("hmm", False, "bool", None),
    ("minSize", 5, "int", None),
    ("modularity", .05, "unit_float_reset", None),
    ("psig", .05, "unit_float_keep", None),
    ("convergenceRounds", 5, "int", None),
    ("lookAhead", .2, "lookahead", None),
    ("louvainRounds", 20, "int", None),
    ("chromosomeGroupFile", '', "str", "saveFilesDirectory"),
    ("chromosomeOrderFile", '', "str", "saveFilesDirectory"),
    ("chromosomePlotSuffix", '', "str", None),
    ("fullGenomePlot", '', "str", "savePlotsDirectory"),
    ("fullGenomePlotTitle", '', "str", None),
    ("plotOrderFile", '', "str", "saveFilesDirectory"),
    ("nScaffolds", 6, "int", None),
    ("scanScaffolds", 5, "int", None),
    ("lengthCutoff", 500000, "int", None),
    ("restrictionSiteFile", '', "str", None),
    ("validPairFile", '', "str", None),
    ("finalOrderingsFile", '', "str", "saveFilesDirectory"),
    ("originalFastaFile", '', "str", None),
    ("assembledFastaFile", '', "str", "saveFilesDirectory"),
]
_BY_KEY = {k: (kind, prefix) for k, _d, kind, prefix in _SPEC}


def _convert(values, key, text):
    kind, prefix = _BY_KEY[key]
    if kind == "str":
        values[key] = (values[prefix] + '/' + text) if prefix else text
    elif kind == "int_required":
        try:
            values[key] = int(text)
        except Exception:
            print("ERROR... resolution must be a an integer value equal to the resolution of the contact map used. Exiting...")
            sys.exit()
    elif kind == "bool":
        if text in ("True", "true"):
            values[key] = True
        elif text in ("False", "false"):
            values[key] = False
    elif kind == "int":
        try:
            values[key] = int(text)
        except Exception:
            print("WARNING... {0} must be an integer value... keeping {0}={1}".format(key, values[key]))
    elif kind == "unit_float_reset":          # modularity: out of range -> default (RUN:123-131)
        try:
            v = float(text)
            if v > 1.:
                print("WARNING... {} must be a value between 0.0 and 1.0... using .05".format(key))
                v = .05
            values[key] = v
        except Exception:
            print("WARNING... {0} must be a floating point value... keeping {0}={1}".format(key, values[key]))
    elif kind == "unit_float_keep":           # psig: out of range -> leave as is (RUN:141-149)
        try:
            v = float(text)
            if v > 1.:
                print("WARNING... {} must be a value between 0.0 and 1.0... keeping {}".format(key, values[key]))
            else:
                values[key] = v
        except Exception:
            print("WARNING... {0} must be a floating point value... keeping {0}={1}".format(key, values[key]))
    elif kind == "lookahead":                 # RUN:160-174
        try:
            v = float(text)
            values[key] = .2 if v > 1. else v
        except Exception:
            values[key] = False if text in ("False", "false") else .2


def readConfigFileToVariables(configFile):
    """RUN:9-219."""
    values = {k: d for k, d, _kind, _p in _SPEC}
    with open(configFile) as fh:
        for raw in fh:
            line = raw.strip('\r').strip('\n')
            if line == '' or line[0] == "#":
                continue
            parts = line.split(' = ')
            if len(parts) < 2:
                print("WARNING... config line without ' = ' skipped (the reference parser raises IndexError here): " + line)
                continue
            key, text = parts[0], parts[1]
            if key in _BY_KEY and text:
                _convert(values, key, text)
    return values


def ensureAllVariablesAreSet(varDict):
    """RUN:221-245: True means "do not run"."""
    unset = [k for k, v in varDict.items() if isinstance(v, str) and v == '']
    if unset:
        print("The following variable(s) do not have any value assossicated with them. Please set this variables to continue.")
        for k in unset:
            print(k)
        print("Exiting...")
        return True
    if varDict["hyperGeom"] is True and varDict["hmm"] is True:
        print('- WARNING - Both hyperGeom and hmm options are set to True... Set one option to "True" and the other '
              'to "False" or both to "False" in order to continue. Exiting...')
        return True
    if varDict["hyperGeom"] is not True:
        # Not in the reference: its hmm = True / both-False paths need hmmlearn's stochastic EM (scaffoldToChromosomes.py:
        # 730-942; SURVEY.md section 2 row 7, out of scope).  Said here, before any part has touched a file, instead
        # of as a NotImplementedError in the middle of Part 1.
        print('- ERROR - this MI355X build implements the hyperGeom = True boundary finder only (hmm = True needs '
              'hmmlearn; see README.md). Set "hyperGeom = True" and "hmm = False" to continue. Exiting...')
        return True
    return False


def _parse_args(argv):
    parser = argparse.ArgumentParser(description="Runs the parts of the HiC assembly pipeline (Parts 1 and 2 on MI355X).")
    for k in (1, 2, 3, 4):
        parser.add_argument("-part%d" % k, help="Run part%d of the pipeline" % k, action='store_true')
    parser.add_argument("-config", help="Full file path to the config file. All arguments must have a value",
                        required=True, type=str)
    parser.add_argument("-device", help="GPU index (default 0)", type=int, default=0)
    return parser.parse_args(argv)


def main(argv=None):
    args = _parse_args(argv)
    t0 = time.time()
    v = readConfigFileToVariables(args.config)
    if ensureAllVariablesAreSet(v):
        sys.exit()
    resident = None
    if args.part1:
        from . import scaffoldToChromosomes as part1
        # -part1 -part2 in one run: the contact matrix stays in HBM for Part 2 (the reference parses the text matrix a
        # second time, OG:688-690; HICMI_REPARSE_FOR_PART2=1 does that too)
        keep = bool(args.part2) and not os.environ.get("HICMI_REPARSE_FOR_PART2")
        resident = part1.runPipeline(v["hicProBedFile"], v["hicProBiasFile"], v["hicProMatrixFile"], v["hicProScaffSizeFile"],
                                     v["dendrogramOrderFile"], v["avgClusterPlot"], v["avgClusterPlot_outlined"],
                                     v["binGroupFile"], v["assessmentFile"], v["chromosomeGroupFile"],
                                     v["hyperGeom"], v["hmm"], v["minSize"], v["modularity"], v["louvainRounds"],
                                     v["psig"], v["convergenceRounds"], v["lookAhead"], v["resolution"], device=args.device,
                                     keep_resident=keep)
    if args.part2:
        from . import orderGenome as part2
        part2.runPipeline(v["hicProBedFile"], v["hicProBiasFile"], v["hicProMatrixFile"], v["chromosomeGroupFile"],
                          v["chromosomeOrderFile"], v["savePlotsDirectory"], v["chromosomePlotSuffix"],
                          v["fullGenomePlot"], v["fullGenomePlotTitle"], v["plotOrderFile"],
                          v["nScaffolds"], v["scanScaffolds"], v["resolution"], device=args.device, resident=resident)
    if args.part3:
        from . import orientSmallScaffolds as part3
        part3.runPipeline(v["chromosomeOrderFile"], v["hicProScaffSizeFile"], v["restrictionSiteFile"], v["validPairFile"],
                          v["finalOrderingsFile"], v["lengthCutoff"], v["resolution"])
    if args.part4:
        from . import writeAssembledFasta as part4
        part4.runPipeline(v["originalFastaFile"], v["finalOrderingsFile"], v["assembledFastaFile"])
    print("Total run-time = " + str(time.time() - t0) + " seconds")


if __name__ == "__main__":
    main()
