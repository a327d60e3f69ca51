for t in 0 3; do
HMCG_SCATTER_THREADS=$t timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); h=j[\"extra\"][\"end_to_end_host_entry\"]; print('threads $t', round(j[\"ms_per_step\"],3), 'py', round(h[\"ms_per_call\"],3), 'lib', round(h[\"library_call_ms\"],3), 'C', h[\"c_caller\"][\"ms_per_call\"], h[\"c_caller\"][\"min_ms\"], h[\"c_caller\"][\"max_ms\"])"
done
