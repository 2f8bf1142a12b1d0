cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
{ echo "== 3 notes, B=64, lr 1e-3"; timeout 400 python3 profiles/tools/r6_tone_learning.py 2000 64 1e-3 2>&1 | grep -v amdgpu.ids
  echo "== 1 note, B=64, lr 1e-3"; MRMT3_TRAJ_NOTES=1 timeout 400 python3 profiles/tools/r6_tone_learning.py 2000 64 1e-3 2>&1 | grep -v amdgpu.ids
  echo "== 1 note, B=16, lr 3e-4"; MRMT3_TRAJ_NOTES=1 timeout 400 python3 profiles/tools/r6_tone_learning.py 2000 16 3e-4 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r6/s4c_tone_learning.txt
