#!/bin/bash
# round 6, call u: WanModel.forward_pair (block-0 prefix shared by the two guidance branches): the new test, the generate / pipeline tests that now go through it,
# the multi-rank bench code path, and a bench line with the shared_prefix leg
TAG=${1:-r06u}
mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -x -k "forward_pair or pipeline or fullsize_generate_call or launcher or bench_multirank or context_cache or usp or ulysses" 2>&1 | tail -8) > gpurun_out/${TAG}_pytest.log
python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench1080p.json.log 2> gpurun_out/${TAG}_bench.err
tail -5 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_bench1080p.json.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','shared_prefix','sec_per_video','sec_per_video_shared_prefix')}); print(d['roofline'])"; tail -3 gpurun_out/${TAG}_bench.err
