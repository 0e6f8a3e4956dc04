for v in 0 24; do echo "== inlane $v"; RTM_FRONT_INLANE=$v timeout -k 10 200 python -u tools/time_split_streams.py 2>&1 | grep front_streams | head -2; done
