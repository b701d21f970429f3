"""tools/lane_log.py (offline costing of ray regroupings from the diagnostic build's step log) on a hand-made log with known answers."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("lane_log", os.path.join(ROOT, "tools", "lane_log.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def _entry(wave, off, kind, lanes):
    m = 0
    for l in lanes:
        m |= 1 << l
    return (wave, off | kind, m & 0xFFFFFFFF, m >> 32)


def test_known_log(tmp_path):
    ll = _load()
    END = 0x1000
    rows = []
    # wave 0: every lane walks records 0x00 and 0x40; lane 3 alone then walks 0x80, 0xC0 (a triangle leaf), 0x150: a single-lane run of 3
    rows += [_entry(0, 0x00, 0, range(64)), _entry(0, 0x40, 0, range(64)), _entry(0, 0x80, 0, [3]), _entry(0, 0xC0, 1, [3]), _entry(0, 0x150, 0, [3])]
    # wave 1: the same two shared records, then lane 60 alone at 0x80 and 0xC0 -- the records wave 0's lane 3 visits
    rows += [_entry(1, 0x00, 0, range(64)), _entry(1, 0x40, 0, range(64)), _entry(1, 0x80, 0, [60]), _entry(1, 0xC0, 1, [60])]
    rows += [_entry(0, END, 2, range(64)), _entry(1, END, 2, range(64))]          # the END record is left out of every figure
    np.array(rows, dtype=np.uint32).tofile(tmp_path / "launch_000_p_128.bin")
    r = ll.analyse(str(tmp_path / "launch_000_p_128.bin"))
    assert r["steps"] == 9 and r["sampled_waves"] == 2
    assert abs(r["lane_use"] - (4 * 64 + 5) / (64.0 * 9)) < 1e-12
    assert r["steps_by_lanes"]["1"] == round(5 / 9, 4) and r["steps_by_lanes"]["33-64"] == round(4 / 9, 4)
    # ray-stream bound: alone, each wave needs its own steps; together, 0x80 and 0xC0 are served once (2 rays), the shared records twice (128 rays)
    assert r["ray_stream_bound_steps_ratio_by_block_waves"]["1"] == 1.0
    assert r["ray_stream_bound_steps_ratio_by_block_waves"]["4"] == round((2 + 2 + 1 + 1 + 1) / 9, 4)
    # single-lane runs: one of 3 steps (wave 0) and one of 2 (wave 1)
    runs = r["single_lane_runs_by_min_length"]
    assert runs["1"]["runs"] == 2 and runs["1"]["share_of_steps"] == round(5 / 9, 4)
    assert runs["2"]["runs"] == 2 and runs["4"]["runs"] == 0
    assert r["kind_share_single_lane_steps"]["tri_leaf"] == round(2 / 5, 4)
    # tail compaction at T = 4: both waves are cut when one ray is left; the two left-overs merge into ONE wave that walks the union {0x80, 0xC0, 0x150}
    assert r["tail_compaction_steps_ratio_by_threshold"]["4"] == round((2 + 2 + 3) / 9, 4)
    # one stage = today's walk
    assert r["staged"]["2"]["steps"] <= 1.0 + 1e-9
