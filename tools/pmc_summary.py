"""rocprofv3 --pmc counter_collection CSVs -> {kernel: {counter: {launches, avg_KB}}} (profiles/*_pmc_hbm_traffic.json).

    python tools/pmc_summary.py OUT.json DIR_WITH_FETCH_SIZE_PASS DIR_WITH_WRITE_SIZE_PASS

FETCH_SIZE and WRITE_SIZE are collected in separate passes (they do not fit one, MI355X_MICROARCH.md); values are KB
per dispatch as rocprofv3 reports them -- bench.py applies the gfx950 correction (FETCH_SIZE x 2 for 16-byte loads)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).strip()
    return name.replace('void ', '')


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = res.setdefault(short(row['Kernel_Name']), {}).setdefault(row['Counter_Name'], [0, 0.0])
                    k[0] += 1
                    k[1] += float(row['Counter_Value'])
    res = {kn: {c: dict(launches=v[0], avg_KB=v[1] / v[0]) for c, v in cs.items()} for kn, cs in res.items()}
    with open(out, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for kn in sorted(res):
        if 'kernel' in kn and 'at::' not in kn:
            print(kn, {c: round(v['avg_KB'], 1) for c, v in res[kn].items()})


if __name__ == '__main__':
    main()
