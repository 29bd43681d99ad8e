"""GGF side-output of self-play (reversi_zero/lib/ggf.py:35-100): move <-> action naming and the
game string.  Note the reference's naming is transposed w.r.t. standard Othello: the LETTER is the
row and the DIGIT the column ("A1" = 0, "F5" = 44, test/lib/test_ggf.py:32-43)."""
from datetime import datetime


def convert_move_to_action(move_str):
    if move_str[:2].lower() == "pa":
        return None
    pos = move_str.lower()
    return (ord(pos[0]) - ord("a")) * 8 + int(pos[1]) - 1


def convert_action_to_move(action):
    if action is None:
        return "PA"
    return chr(ord("A") + action // 8) + str(action % 8 + 1)


def make_ggf_string(black_name=None, white_name=None, dt=None, moves=None, result=None, think_time_sec=60):
    dt = dt or datetime.utcnow()
    move_list = "".join(("B[%s]" if i % 2 == 0 else "W[%s]") % m for i, m in enumerate(moves or []))
    return ("(;GM[Othello]PC[RAZSelf]DT[%s]PB[%s]PW[%s]RE[%s]TI[%s]TY[8]"
            "BO[8 ---------------------------O*------*O--------------------------- *]%s;)") % (
        dt.strftime("%Y.%m.%d_%H:%M:%S.%Z"), black_name or "black", white_name or "white", result or "?",
        f"{think_time_sec // 60}:{think_time_sec % 60}", move_list)
