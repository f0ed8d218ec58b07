for f in columns rows columns rows; do
NAF_FUSE=$f timeout -k 10 200 python benchmarks/host_api_steps.py 2>&1 | tail -1 | cut -c1-200
done
