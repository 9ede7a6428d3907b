"""numpy.random.Generator look-alikes on the engine's counter-based streams.

ORACLE / TEST INFRASTRUCTURE ONLY.

The reference threads ONE ``np.random.Generator`` through every actor
(simulation_runner.py:77,138,177,237).  Its own tests replace that object with
fakes exposing the same method names (tests/unit/runtime/actors/test_edge.py:31-49,
tests/integration/single_server/test_int_single_server.py:36).  The classes
below use that seam to make the UNMODIFIED reference actors draw from the
spec in oracle/oracle_rng.h (one stream per actor, logical draw index = number
of draws that actor made so far), so that reference, C oracle and HIP engine
can be compared bit for bit.

Arithmetic is done by the C functions of libaf_oracle.so (via ctypes), so the
Python reference and the C oracle share every rounding.
"""

from __future__ import annotations

from . import oracle_lib as ol

_DIST = {"poisson": 0, "normal": 1, "log_normal": 2, "exponential": 3, "uniform": 4}


class GeneratorRNG:
    """Stream 0: users-per-window draws and gap uniforms (samplers/*.py)."""

    def __init__(self, seed: int) -> None:
        self.seed, self.stream, self.n = int(seed), ol.STREAM_GENERATOR, 0
        self._L = ol.lib()

    def _next(self) -> int:
        i = self.n
        self.n += 1
        return i

    def poisson(self, lam: float) -> int:  # poisson_poisson.py:60
        return int(self._L.orc_x_poisson(float(lam), self.seed, self.stream, self._next(), 0))

    def normal(self, mean: float, sigma: float) -> float:  # gaussian_poisson.py:72-76
        u = self._L.orc_x_uniform(self.seed, self.stream, self._next(), 0)
        return float(mean) + float(sigma) * self._L.orc_x_norminv(u)

    def random(self) -> float:  # poisson_poisson.py:69
        return self._L.orc_x_uniform(self.seed, self.stream, self._next(), 0)


class EdgeRNG:
    """Stream 1+e: per send, uniform 0 = dropout draw, uniforms 1.. = latency."""

    def __init__(self, seed: int, edge_index: int) -> None:
        self.seed, self.stream = int(seed), ol.stream_edge(edge_index)
        self.sends = 0      # logical draw index of the NEXT send
        self.latencies = 0  # sends that survived the dropout test
        self._cur = -1
        self._L = ol.lib()

    def uniform(self) -> float:  # edge.py:78
        self._cur = self.sends
        self.sends += 1
        return self._L.orc_x_uniform(self.seed, self.stream, self._cur, 0)

    def _variate(self, dist: str, mean: float, sigma: float) -> float:
        self.latencies += 1
        return self._L.orc_x_variate(_DIST[dist], float(mean), float(sigma), self.seed, self.stream, self._cur, 1)

    # general_sampler (samplers/common_helpers.py:49-89) dispatches to these:
    def random(self) -> float:
        return self._variate("uniform", 0.0, 0.0)

    def poisson(self, lam: float) -> int:
        return int(self._variate("poisson", lam, 0.0))

    def exponential(self, scale: float) -> float:
        return self._variate("exponential", scale, 0.0)

    def normal(self, mean: float, sigma: float) -> float:
        # un-truncated; truncated_gaussian_generator applies max(0, .) itself
        self.latencies += 1
        u = self._L.orc_x_uniform(self.seed, self.stream, self._cur, 1)
        return float(mean) + float(sigma) * self._L.orc_x_norminv(u)

    def lognormal(self, mean: float, sigma: float) -> float:
        return self._variate("log_normal", mean, sigma)


class ServerRNG:
    """Stream 0x1000+s: endpoint pick per request arriving at server s."""

    def __init__(self, seed: int, server_index: int) -> None:
        self.seed, self.stream, self.n = int(seed), ol.stream_server(server_index), 0
        self._L = ol.lib()

    def integers(self, low: int = 0, high: int | None = None) -> int:  # server.py:101
        if high is None:
            low, high = 0, low
        span = int(high) - int(low)
        if span <= 0:
            msg = "low >= high"
            raise ValueError(msg)  # numpy raises the same for integers(0, 0)
        i = self.n
        self.n += 1
        if span == 1:
            return int(low)
        return int(low) + ((self._L.orc_x_word0(self.seed, self.stream, i) * span) >> 32)
