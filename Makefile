# Convenience targets; the real build is snappy_amd/csrc/Makefile (hipcc, gfx950) and oracle/Makefile (gcc).
all:
	$(MAKE) -C snappy_amd/csrc
	$(MAKE) -C oracle

test: all          # CPU suite: oracle vs golden vectors, host logic, ABI surface, lane simulator, gloo sharding
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all      # needs an MI355X: bit-exact parity through the C ABI
	python -m pytest tests -x -q -m gpu

bench: all
	python bench.py

clean:
	$(MAKE) -C snappy_amd/csrc clean
	$(MAKE) -C oracle clean

.PHONY: all test test-gpu bench clean
