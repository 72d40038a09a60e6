# A/B of two builds of libgoldsrl.so on one box, alternating: bash tools/ab_lib.sh OTHER.so ROUNDS [bench flags...]   (on the GPU box)
# prints ms_per_step of bench.py for the tree's library (A) and OTHER (B), ROUNDS times each
set -e
OTHER="$1"; ROUNDS="$2"; shift 2
cd "$GRAFT_REPO_ROOT"
L=golds-rl-gym_amd/lib
cp $L/libgoldsrl.so $L/_a.so; cp "$OTHER" $L/_b.so
for r in $(seq 1 $ROUNDS); do
  for v in a b; do
    cp $L/_$v.so $L/libgoldsrl.so
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --no-legs --no-shard --no-flat-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), flush=True)"
  done
done
cp $L/_a.so $L/libgoldsrl.so; rm -f $L/_a.so $L/_b.so
