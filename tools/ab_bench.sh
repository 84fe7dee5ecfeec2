#!/bin/bash
# one headline line + one streaming line, key numbers only (used under tools/gpu_ab_lib.sh)
python bench.py --steps ${AB_STEPS:-3} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value', 'decode_ms_per_token', 'encode_ms', 'prefill_ms')}, 'mismatches', d['parity']['mismatches'])"
python bench.py --mode stream --seconds ${AB_SECONDS:-60} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('stream', d['value'], d.get('chunk_latency_ms'))"
