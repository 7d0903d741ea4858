#!/usr/bin/env python3
"""Content hash of the kernel sources (bayes_sim_ig_amd/csrc/* + include/bsig.h): what a profile
is a profile OF.  tools/round_profiles.sh / final_round.sh stamp it into every file they write
("# csrc: <hash>"); bench.pmc_traffic compares it with the tree it runs in, with or without git."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, 'bayes_sim_ig_amd', 'csrc', '*')))
    files.append(os.path.join(root, 'include', 'bsig.h'))
    for path in files:
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_hash())
