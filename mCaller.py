#!/usr/bin/env python3
"""Entry point with the reference's name: `./mCaller.py -m GATC -r ref.fasta -d model.pkl -e reads.eventalign.tsv -f reads.fastq`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mcaller_amd.mCaller import main

if __name__ == '__main__':
    main()
