#!/bin/bash
# which earlier test file leaves the process in a state in which tests/test_gpu_parity.py::test_create_destroy_cycles_do_not_leak crashes
cd "$GRAFT_REPO_ROOT"
T=tests/test_gpu_parity.py::test_create_destroy_cycles_do_not_leak
for f in "$@"; do
  env $BISECT_ENV python -m pytest $f $T -q -m gpu -p no:cacheprovider > /tmp/bis.log 2>&1; rc=$?
  echo "$f -> rc $rc $(grep -c 'Fatal Python' /tmp/bis.log) fatal; $(grep -E 'passed|failed' /tmp/bis.log | tail -1)"
  [ $rc -ne 0 ] && grep -B2 -A12 "Fatal Python\|^E " /tmp/bis.log | grep -v "site-packages\|dist-packages" | head -30
done
