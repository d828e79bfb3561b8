#!/usr/bin/env python3
"""Runs one pytest selection N times in ONE process and prints how often it failed (a check for run-to-run
differences: atomics, stream races).

    python tools/repeat_test.py N <pytest args...>
"""
import sys

import pytest

n = int(sys.argv[1])
bad = 0
for i in range(n):
    rc = pytest.main(["-x", "-q", "--timeout", "120", "--timeout-method", "thread", "-p", "no:cacheprovider"] + sys.argv[2:])
    print(f"[repeat {i}] rc={int(rc)}", flush=True)
    bad += int(rc) != 0
print(f"{bad} of {n} runs failed")
sys.exit(1 if bad else 0)
