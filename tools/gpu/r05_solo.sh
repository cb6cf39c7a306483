for n in 747 3000 100000; do
timeout 300 python tools/step_probe.py --n $n --p 25 --trees 75 --sweeps 20 --iters 60 --path persistent 2>&1 | tail -1 | cut -c1-400
done
