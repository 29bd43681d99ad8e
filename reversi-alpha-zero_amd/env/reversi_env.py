"""Drop-in for reversi_zero/env/reversi_env.py: ReversiEnv / Board / Player / Winner with the
reference's attribute names, return values and error behaviour (env/reversi_env.py:9-143), the
state transition itself executed by libraz (`raz_env_step`, include/raz.h)."""
import ctypes
import enum
from logging import getLogger

from .._native import lib, check
from ..lib.bitboard import board_to_string, bit_count

logger = getLogger(__name__)
# noinspection PyArgumentList
Player = enum.Enum("Player", "black white")
# noinspection PyArgumentList
Winner = enum.Enum("Winner", "black white draw")


def another_player(player: Player):
    return Player.white if player == Player.black else Player.black


class ReversiEnv:
    def __init__(self):
        self.board = None
        self.next_player = None  # type: Player
        self.turn = 0
        self.done = False
        self.winner = None  # type: Winner

    def reset(self):
        self.board = Board()
        self.next_player = Player.black
        self.turn = 0
        self.done = False
        self.winner = None
        return self

    def update(self, black, white, next_player):
        self.board = Board(black, white)
        self.next_player = next_player
        self.turn = sum(self.board.number_of_black_and_white) - 4
        self.done = False
        self.winner = None
        return self

    def step(self, action):
        """action: 0..63 (0 = top left, 63 = bottom right) or None to resign.
        Returns (board, {}) like the reference (env/reversi_env.py:42-74)."""
        assert action is None or 0 <= action <= 63, f"Illegal action={action}"
        b = ctypes.c_uint64(self.board.black)
        w = ctypes.c_uint64(self.board.white)
        p = ctypes.c_uint8(self.next_player.value)
        s = ctypes.c_uint8(0)
        legal = ctypes.c_uint64(0)
        check(lib.raz_env_step(ctypes.byref(b), ctypes.byref(w), ctypes.byref(p), ctypes.byref(s),
                               ctypes.byref(legal), 255 if action is None else int(action)),
              "raz_env_step")
        st = s.value
        if st & 0x10:  # env/reversi_env.py:90-93
            logger.warning(f"Illegal action={action}, No Flipped!")
        if not (st & 0x30):  # a disc was placed
            self.board.black, self.board.white = b.value, w.value
            self.turn += 1
        self.next_player = Player(p.value)
        if st & 0x0f:
            self.done = True
            if self.winner is None:  # env/reversi_env.py:78
                self.winner = Winner(st & 0x0f)
        return self.board, {}

    def get_own_and_enemy(self):
        if self.next_player == Player.black:
            return self.board.black, self.board.white
        return self.board.white, self.board.black

    def set_own_and_enemy(self, own, enemy):
        if self.next_player == Player.black:
            self.board.black, self.board.white = own, enemy
        else:
            self.board.white, self.board.black = own, enemy

    def change_to_next_player(self):
        self.next_player = another_player(self.next_player)

    def render(self):
        b, w = self.board.number_of_black_and_white
        print(f"next={self.next_player.name} turn={self.turn} B={b} W={w}")
        print(board_to_string(self.board.black, self.board.white, with_edge=True))

    @property
    def observation(self):
        return self.board


class Board:
    def __init__(self, black=None, white=None, init_type=0):
        # `x or default`: a 0 bitboard is replaced by the initial pattern (env/reversi_env.py:135-136)
        self.black = black or (0b00010000 << 24 | 0b00001000 << 32)
        self.white = white or (0b00001000 << 24 | 0b00010000 << 32)
        if init_type:
            self.black, self.white = self.white, self.black

    @property
    def number_of_black_and_white(self):
        return bit_count(self.black), bit_count(self.white)
