#!/bin/bash
for lib in _build_v0/lib.so dbgphmm_amd/libphmm_amd.so; do
  echo "== $lib"; PHMM_AMD_LIB=$PWD/$lib timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | grep "shard of" | tail -3
done
echo "== current, PHMM_NO_KEEP_ALL=1"; PHMM_NO_KEEP_ALL=1 timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | grep "shard of" | tail -3
for n in 2 4; do echo "== current, shard of $n"; timeout -k 10 300 python tools/trace_shard.py $n 2>&1 | grep "shard of" | tail -2; done
echo "== trace"; PHMM_NO_KEEP_ALL=1 PHMM_TRACE=1 timeout -k 10 300 python tools/trace_shard.py 8 2>&1 | tail -42
