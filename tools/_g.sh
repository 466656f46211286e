python tools/dense_t16_bench.py waves=0,8,12
python tools/dense_small.py shape=2449029,100,100 13=3
python tools/dense_small.py shape=2449029,100,128 13=3
python tools/gcn_layer_one.py
