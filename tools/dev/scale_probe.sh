#!/bin/bash
# A/B of the working tree against another commit ON ONE GPU BOX (box-to-box spread is +-3 %, more than most changes):
#   git worktree add /tmp/base/wt <commit>; (cd /tmp/base/wt && python -m uforecon_amd.build)
#   mkdir _old_tree && (cd /tmp/base/wt && tar cf - --exclude=.git .) | (cd _old_tree && tar xf -)      # _old_tree/ is git-ignored
#   gpurun -- 'bash tools/dev/scale_probe.sh > gpurun_out/scale_probe.txt 2>&1'
# Prints ms per frame and the single-stream kernel split for configs[1] and configs[3], alternating the two trees.
run() { python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms_per_frame_rank0']; print('$1', round(d['ms_per_step'],1), round(k['view_transformer'],1), round(k['ray_transformer'],1), round(k['gather'],1))"; }
for ARGS in "" "--height 600 --width 800 --views 5 --coarse 128 --fine 128"; do
  run NEW
  [ -d _old_tree ] && (cd _old_tree && run OLD)
  run NEW
  [ -d _old_tree ] && (cd _old_tree && run OLD)
done
