"""ref_harness.py — import the UNMODIFIED reference (mokemokechicken/reversi-alpha-zero) from
/root/reference/src inside THIS container, for golden-vector generation and oracle pinning.

TEST INFRASTRUCTURE ONLY.  Nothing under reversi-alpha-zero_amd/ may import this module, and it
must never be needed at run time on the GPU box (where /root/reference does not exist): its only
consumers are tests/golden/make_golden*.py (run here, outputs committed under tests/golden/), CPU tests marked
`needs_reference` (auto-skipped when the reference tree is absent) and bench.py's `cpu_baseline` leg, which times the
reference's own self-play on the bench box's host cores from the byte-compiled copy oracle/build_ref.py stages
under oracle/_ref/ (build output, git-ignored).

No reference file is edited or copied.  What is shimmed, and why:
  * keras.* / tensorflow: un-vendored third-party deps of agent/model.py, agent/api.py
    (requirements.txt:25,59) -> MagicMock modules; the net arithmetic is restated separately;
  * moke_config (requirements.txt:31): `ConfigBase` marker + `create_config(cls, dict)` recursive
    attribute overlay (manager.py:43-45 is its only call site);
  * nose (test/ uses nose.tools eq_/ok_) -> three-line shim so the reference tests run under pytest;
  * `with await self.sem` (agent/player.py:205) is a py<=3.8 idiom: ReversiPlayer.sem is replaced,
    per instance, by a Semaphore whose __await__ acquires and returns a releasing context manager;
  * asyncio.get_event_loop() at player.py:53 needs a current loop on py3.10+: one is installed.
"""
import asyncio
import os
import sys
import types
from unittest.mock import MagicMock

def _reference_root():
    """/root/reference where it exists (the build container); else oracle/_ref - the same modules byte-compiled from
    there by oracle/build_ref.py (git-ignored build output that travels to the GPU box, where bench.py's cpu_baseline
    leg times the reference's own self-play on the host cores).  RAZ_REFERENCE_ROOT overrides."""
    env = os.environ.get("RAZ_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/src/reversi_zero"):
        return "/root/reference"
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


REFERENCE_ROOT = _reference_root()
REFERENCE_SRC = os.path.join(REFERENCE_ROOT, "src")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "reversi_zero"))


_installed = False


def _overlay(obj, d):
    for k, v in d.items():
        cur = getattr(obj, k, None)
        if isinstance(v, dict) and cur is not None and not isinstance(cur, (dict, list, tuple, int, float, str)):
            _overlay(cur, v)
        else:
            setattr(obj, k, v)
    return obj


def install():
    """Put the shims and the reference on sys.path / sys.modules (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_SRC}")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    for m in ["keras", "keras.backend", "keras.engine", "keras.engine.topology",
              "keras.engine.training", "keras.layers", "keras.layers.convolutional",
              "keras.layers.core", "keras.layers.merge", "keras.layers.normalization",
              "keras.losses", "keras.regularizers", "keras.callbacks", "keras.optimizers",
              "tensorflow", "dotenv"]:
        sys.modules.setdefault(m, MagicMock())

    mc = types.ModuleType("moke_config")

    class ConfigBase:
        pass

    def create_config(cls, d=None):
        return _overlay(cls(), d or {})

    mc.ConfigBase = ConfigBase
    mc.create_config = create_config
    sys.modules.setdefault("moke_config", mc)

    nose = types.ModuleType("nose")
    tools = types.ModuleType("nose.tools")
    trivial = types.ModuleType("nose.tools.trivial")

    def eq_(a, b, msg=None):
        assert a == b, msg or f"{a!r} != {b!r}"

    def ok_(x, msg=None):
        assert x, msg

    def assert_almost_equal(a, b, places=7, msg=None):
        assert round(abs(a - b), places) == 0, msg or f"{a!r} !~ {b!r}"

    for mod in (tools, trivial):
        mod.eq_, mod.ok_, mod.assert_almost_equal = eq_, ok_, assert_almost_equal
    nose.tools = tools
    tools.trivial = trivial
    sys.modules.setdefault("nose", nose)
    sys.modules.setdefault("nose.tools", tools)
    sys.modules.setdefault("nose.tools.trivial", trivial)

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    try:
        asyncio.get_event_loop_policy().get_event_loop()
    except RuntimeError:
        asyncio.set_event_loop(asyncio.new_event_loop())
    _installed = True


class _FifoTimerHandle(asyncio.TimerHandle):
    """TimerHandle ordered by (deadline, scheduling sequence): equal deadlines fire first-set-first."""
    __slots__ = ("_seq",)

    def __lt__(self, other):
        return (self._when, self._seq) < (other._when, other._seq)


class VirtualTimeLoop(asyncio.SelectorEventLoop):
    """asyncio event loop in EXACT VIRTUAL TIME, the deterministic stage on which the unmodified
    reference player is run with parallel_search_num > 1 (raz-sched-v1, oracle/orc_mcts.c):
      * time() is a Fraction that never moves while a callback is runnable; when nothing is
        runnable it jumps to the earliest timer (so computation takes no time and
        prediction_worker_sleep_sec / wait_for_expanding_sleep_sec only order events);
      * delays are taken as exact decimals (1e-05 is 1/100000), so ten 10-us sleeps end exactly
        when one 100-us sleep does, and equal deadlines fire in the order they were set.
    Nothing of the reference is edited: ReversiPlayer.__init__ picks the loop up through
    asyncio.get_event_loop() (agent/player.py:53)."""

    def __init__(self):
        super().__init__()
        from fractions import Fraction
        self._F = Fraction
        self._vnow = Fraction(0)
        self._vseq = 0

    def time(self):
        return self._vnow

    def call_later(self, delay, callback, *args, context=None):
        return self.call_at(self._vnow + self._F(repr(float(delay))), callback, *args, context=context)

    def call_at(self, when, callback, *args, context=None):
        import heapq
        self._check_closed()
        timer = _FifoTimerHandle(when, callback, args, self, context)
        timer._seq = self._vseq
        self._vseq += 1
        heapq.heappush(self._scheduled, timer)
        timer._scheduled = True
        return timer

    def _run_once(self):
        import heapq
        if not self._ready:
            while self._scheduled and self._scheduled[0]._cancelled:
                self._timer_cancelled_count -= 1
                h = heapq.heappop(self._scheduled)
                h._scheduled = False
            if self._scheduled and self._scheduled[0]._when > self._vnow:
                self._vnow = self._scheduled[0]._when
        super()._run_once()


class _Releaser:
    def __init__(self, sem):
        self._sem = sem

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        self._sem.release()


class CompatSemaphore(asyncio.Semaphore):
    """`with await sem:` support (agent/player.py:205) on Python >= 3.9."""

    def __await__(self):
        yield from self.acquire().__await__()
        return _Releaser(self)


def load_config(yml_name=None, overrides=None):
    """Config() overlaid with config/<yml_name> (manager.py:41-45 semantics) and `overrides`."""
    install()
    import yaml
    from reversi_zero.config import Config
    cfg = Config()
    if yml_name:
        with open(os.path.join(REFERENCE_ROOT, "config", yml_name), "rt") as f:
            _overlay(cfg, yaml.safe_load(f))
    if overrides:
        _overlay(cfg, overrides)
    return cfg


def make_player(config, api, enable_resign=True, mtcs_info=None):
    """ReversiPlayer(config, None, ...) with the py3.10 semaphore shim applied."""
    install()
    from reversi_zero.agent.player import ReversiPlayer
    try:
        asyncio.get_event_loop()
    except RuntimeError:
        asyncio.set_event_loop(asyncio.new_event_loop())
    p = ReversiPlayer(config, None, enable_resign=enable_resign, mtcs_info=mtcs_info, api=api)
    p.sem = CompatSemaphore(p.play_config.parallel_search_num)
    return p
