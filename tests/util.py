import os

import numpy as np

GOLDEN_ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(GOLDEN_ROOT, "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def linf_peak(a, b):
    """Per batch item: max|a-b| / max|b| (the 'L-inf / peak' metric of SURVEY.md / BASELINE.md)."""
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-300)


def rel_l2(a, b):
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return np.sqrt(((a - b) ** 2).sum(1) / np.maximum((b ** 2).sum(1), 1e-300))


def record(test, **values):
    """Print the measured errors of a parity test and append them to gpurun_out/parity_measured.jsonl (merged back from the GPU box;
    the numbers DESIGN.md section 5 quotes and the bounds in the tests are set from these lines, not guessed)."""
    import json
    row = {"test": test}
    for k, v in values.items():
        row[k] = [float(f"{float(e):.3e}") for e in np.ravel(v)] if np.ndim(v) else float(f"{float(v):.3e}")
    line = json.dumps(row)
    print("MEASURED " + line)
    try:
        out = os.path.join(os.path.dirname(GOLDEN_ROOT), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
