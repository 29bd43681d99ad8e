"""reversi-alpha-zero_amd — MI355X-native self-play hot path behind the ReversiEnv / ReversiPlayer /
play_*.json surface of mokemokechicken/reversi-alpha-zero (import name: reversi_alpha_zero_amd).

Layout mirrors the reference package `reversi_zero` for the modules on the path:
    lib/bitboard.py      <- reversi_zero/lib/bitboard.py
    env/reversi_env.py   <- reversi_zero/env/reversi_env.py
    agent/...            <- reversi_zero/agent/{player,api,model}.py
    worker/self_play.py  <- reversi_zero/worker/self_play.py
All compute goes through the C ABI of csrc/libraz.so (include/raz.h); importing `_native` raises
if that library has not been built (`python -c "import __graft_entry__ as g; g.build()"`).
"""
__version__ = "0.1.0"
