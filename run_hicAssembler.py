#!/usr/bin/env python3
"""`python run_hicAssembler.py -part1 -part2 -config FILE` - same command line as the reference driver."""
from hic_genome_assembler_amd.run_hicAssembler import main

if __name__ == "__main__":
    main()
