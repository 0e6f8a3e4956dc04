set -o pipefail
for v in 0 1 8 16 24 32 40 48 56 64; do
  echo "== inlane $v"; RTM_FRONT_INLANE=$v timeout -k 10 100 python -u tools/time_scenes.py 2>&1 | grep -v amdgpu || exit 1
done
