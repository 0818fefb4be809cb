#!/bin/bash
# round-6 GPU session 43: (a) tools/probe_unclosed_session.py - engines left open with a pipeline-owned live session (what a failing test leaves behind), long-kernel engines created
# behind them: does that abort?  (b) the whole suite twice with the tests' fills synchronised before other streams write into the buffers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r6s43
mkdir -p $O
AMD_LOG_LEVEL=1 timeout 600 python tools/probe_unclosed_session.py 12 > $O/probe.log 2>&1; echo "probe rc=$?"; tail -6 $O/probe.log | cut -c1-300
for i in 1 2; do timeout 1200 python -m pytest tests -q -m gpu -x -rf -p no:cacheprovider > $O/all_$i.log 2>&1; echo "all rc=$?"; grep -E "passed|failed" $O/all_$i.log | tail -1; done
