python tools/small_configs.py arxiv 15=0
python tools/small_configs.py arxiv 15=1
python tools/small_configs.py arxiv 15=1 3=0
python tools/small_configs.py arxiv 15=0 3=0
python tools/small_configs.py batched 15=0
python tools/small_configs.py batched 15=1
