python tools/gin_gather_bench.py
