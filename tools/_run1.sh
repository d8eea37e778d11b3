set -e
mkdir -p gpurun_out
echo "== c2 direct parts"; for dp in 2 3 4; do CBA_MODEB_DPARTS=$dp EXP_TAG="dparts$dp" python tools/exp_modeb.py c2 c5 2>&1 | grep -v amdgpu; done
echo "== c3q moment variants"; for v in 4 36 3 5; do CBA_MODEB_VARIANT=$v EXP_TAG="variant$v" python tools/exp_modeb.py c3q c4 2>&1 | grep -v amdgpu; done
