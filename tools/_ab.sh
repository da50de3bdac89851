cd /root/repo
timeout 600 python -m pytest tests -x -q -m gpu -k "adamw or optim or trajectory or determinism" 2>&1 | tail -3
TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o step -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > /tmp/prof.txt 2>&1
grep -h "adamw\|task_loss\|max_pool" /tmp/prof/*/step_kernel_stats.csv /tmp/prof/step_kernel_stats.csv 2>/dev/null | cut -c1-60,100-260 | head
tail -1 /tmp/prof.txt | cut -c1-120
