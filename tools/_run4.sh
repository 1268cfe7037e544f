STEPS=80 WARMUP=10 bash tools/ab_variants.sh base free1 free2 free4 base free1 free2 free4
