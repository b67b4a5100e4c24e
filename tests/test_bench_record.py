"""bench.py's result line (benchlib/record.py) on canned records: the driver keeps only the last few kB of stdout, and round 4
lost its whole record to a 20.7 kB line.  No GPU: the builder is pure."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from benchlib import record  # noqa: E402


def canned():
    """Full records of earlier rounds (the 16-21 kB lines that used to be printed), newest last."""
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r04_*_bench.json')) +
                   glob.glob(os.path.join(ROOT, 'profiles', 'r0[5-9]_*_bench_detail.json')))
    return [(os.path.basename(p), json.load(open(p))) for p in paths]


def published_lines():
    """The result lines of round 5 on (profiles/rNN_*_bench.json: what bench.py printed last, as the driver keeps it)."""
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[5-9]_*_bench.json')))


@pytest.mark.parametrize('path', published_lines())
def test_published_result_lines_are_what_the_driver_can_keep(path):
    raw = open(path).read().strip()
    assert '\n' not in raw and len(raw.encode()) < 4096, (path, len(raw))
    rec = json.loads(raw, parse_constant=lambda c: pytest.fail('non-strict JSON constant ' + c))
    for k in record.REQUIRED_KEYS:
        assert k in rec, k
    assert rec['roofline']['target_frac'] >= 0.40 and rec['cpu_baseline']['cpu_model']


@pytest.mark.parametrize('name,detail', canned())
def test_result_line_fits_the_drivers_window(name, detail):
    detail = dict(detail, detail_file='bench_detail.json')
    line = record.result_line(detail)
    assert '\n' not in line and len(line.encode()) < 4096, (name, len(line))
    rec = json.loads(line, parse_constant=lambda c: pytest.fail('non-strict JSON constant ' + c))
    for k in record.REQUIRED_KEYS:
        assert k in rec, k
    assert rec['value'] == pytest.approx(detail['value'], rel=1e-5)
    assert rec['ms_per_step'] == pytest.approx(detail['ms_per_step'], rel=1e-5)
    assert isinstance(rec['config']['workload'], str)
    rf = rec['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'ms_per_launch', 'bytes_per_launch',
              'target_frac', 'target_kernel', 'target_ms_per_launch', 'target_traffic', 'target_frac_at_1e6'):
        assert k in rf, k
    assert rf['frac'] == pytest.approx(rf['achieved'] / rf['peak'], rel=1e-4)
    assert rf['target_frac'] == pytest.approx(detail['roofline_mt6']['frac'], rel=1e-5)
    # scalars only inside the blocks the driver's record keeps
    for blk in ('roofline', 'cpu_baseline', 'legs', 'config'):
        assert all(not isinstance(v, (dict, list)) for v in rec[blk].values()), blk
    cb = rec['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] and cb['sample']
    if 'cpu_model' in detail['cpu_baseline']:
        assert cb['cpu_model'] == detail['cpu_baseline']['cpu_model']
    for k in ('c3_ukf5_ms', 'c4_tpq_ms', 'c5_n201_ms', 'c5_n21_ms', 'theta_us'):
        assert rec['legs'][k] > 0


def test_result_line_is_strict_json_with_nan_and_numpy_scalars():
    np = pytest.importorskip('numpy')
    d = {'metric': 'm', 'value': np.float64(1.5), 'unit': 'u', 'n_gpus': 1, 'steps': np.int64(20), 'warmup': 1,
         'ms_per_step': float('nan'), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
         'data': 'synthetic', 'config': {'workload': 'w' * 500}, 'roofline': {'frac': float('inf'), 'kernel': 'k<SSMQ_F_UNGM_DYN>'}}
    rec = json.loads(record.result_line(d))
    assert rec['ms_per_step'] is None and rec['roofline']['frac'] is None and rec['value'] == 1.5 and rec['steps'] == 20
    assert rec['roofline']['kernel'] == 'k<UNGM_DYN>' and len(rec['config']['workload']) <= 200


def test_result_line_sheds_optional_keys_before_it_overflows(monkeypatch):
    d = json.load(open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r04_*_bench.json')))[-1]))
    big = dict(d, config=dict(d['config'], workload='x' * 190, launcher='y' * 3000))
    full = json.loads(record.result_line(big))
    assert len(full['config']['launcher']) == 200
    monkeypatch.setattr(record, 'MAX_LINE', 2300)
    line = record.result_line(big)
    assert len(line) <= 2300
    rec = json.loads(line)
    assert all(k in rec for k in record.REQUIRED_KEYS) and 'cpu_baseline' in rec and 'target_frac' in rec['roofline']
    assert len(rec.get('legs', {})) < len(full['legs'])
    monkeypatch.setattr(record, 'MAX_LINE', 500)
    with pytest.raises(ValueError):
        record.result_line(big)


def test_detail_file_is_strict_json(tmp_path):
    np = pytest.importorskip('numpy')
    p = tmp_path / 'd.json'
    record.write_detail({'a': np.float64('nan'), 'b': [np.int32(3), (1.0, float('inf'))], 'c': {'d': np.arange(2)[1]}}, str(p))
    got = json.loads(p.read_text(), parse_constant=lambda c: pytest.fail(c))
    assert got == {'a': None, 'b': [3, [1.0, None]], 'c': {'d': 1}}
