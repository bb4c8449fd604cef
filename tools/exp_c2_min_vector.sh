cd $GRAFT_REPO_ROOT
for V in 8192 1024; do
  for ARGS in "--steps 20" "--method adam --moment1 explicit --steps 20" "--method adam-autograd --steps 20" "--method adam --steps 20"; do
    echo "## MIN_VECTOR_MIB=$V bench.py --no-cpu $ARGS"
    DFX_ARENA_MIN_VECTOR_MIB=$V timeout -k 10 200 python3 bench.py --no-cpu $ARGS 2>>gpurun_out/exp_c2.err || echo FAILED
  done
done
