#!/bin/bash
# The GPU parity tests in reverse collection order (order-dependent state: workspaces, launch graphs, tuning knobs).
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | tac > /tmp/ids.txt
python -m pytest $(cat /tmp/ids.txt | tr '\n' ' ') -x -q -p no:cacheprovider 2>&1 | tail -3
