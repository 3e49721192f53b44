"""Minimal key-value logger with the call surface train.py uses (baselines/logger.py:190-373): info/warn,
record_tabular/dump_tabular -> stdout table + progress.csv (same column protocol as the reference's CSV writer,
logger.py:101-135), configure/get_dir.  The rest of the reference logger (tensorboard, json, ProfileKV) is out of
scope."""
import csv
import os
import sys

_state = {'dir': None, 'kv': {}, 'keys': [], 'csv': None}


def configure(dir=None, resume_epoch=None):
    """resume_epoch (a job that goes on from a checkpoint, curious_amd.checkpoint): progress.csv of `dir` is continued --
    its header stays, its rows behind that epoch (logged by the earlier job between its last checkpoint and its end) go."""
    os.makedirs(dir, exist_ok=True)
    _state['dir'] = dir
    _state['csv'] = None
    _state['keys'] = []
    path = os.path.join(dir, 'progress.csv')
    if resume_epoch is not None and os.path.exists(path):
        with open(path, newline='') as f:
            reader = csv.DictReader(f)
            keys = list(reader.fieldnames or [])
            rows = [r for r in reader if int(float(r.get('epoch', -1))) <= resume_epoch]
        if keys:
            _state['keys'] = keys
            with open(path, 'w', newline='') as f:
                w = csv.DictWriter(f, keys)
                w.writeheader()
                w.writerows(rows)


def get_dir():
    return _state['dir']


def info(*args):
    print(*args, file=sys.stdout, flush=True)


def warn(*args):
    print(*args, file=sys.stderr, flush=True)


warning = warn


def record_tabular(key, val):
    _state['kv'][key] = val


def dump_tabular():
    kv = _state['kv']
    if not kv:
        return
    width = max(len(k) for k in kv)
    print('-' * (width + 20))
    for k in sorted(kv):
        print('| %-*s | %-12s |' % (width, k, kv[k]))
    print('-' * (width + 20), flush=True)
    if _state['dir'] is not None:
        path = os.path.join(_state['dir'], 'progress.csv')
        new_keys = [k for k in sorted(kv) if k not in _state['keys']]
        if new_keys and _state['keys']:
            # header grew: rewrite the file with the extended header (what logger.py:113-126 does)
            rows = list(csv.DictReader(open(path)))
            _state['keys'] += new_keys
            with open(path, 'w', newline='') as f:
                w = csv.DictWriter(f, _state['keys'])
                w.writeheader()
                w.writerows(rows)
        elif new_keys:
            _state['keys'] = new_keys
            with open(path, 'w', newline='') as f:
                csv.DictWriter(f, _state['keys']).writeheader()
        with open(path, 'a', newline='') as f:
            csv.DictWriter(f, _state['keys']).writerow({k: kv.get(k, '') for k in _state['keys']})
    kv.clear()
