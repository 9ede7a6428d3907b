"""SimPy-4.1.1-compatible discrete-event kernel -- ORACLE / TEST INFRASTRUCTURE ONLY.

The reference (AsyncFlow v0.1.1) pins ``simpy = "^4.1.1"``
(/root/reference/pyproject.toml:43, poetry.lock:1234-1235).  SimPy is a third
party dependency that is neither vendored under /root/reference nor installable
in this image (no network, no wheel).  This module restates SimPy's *published*
scheduling algorithm so that the unmodified reference actors can be executed on
the CPU to produce golden vectors for the MI355X engine:

* event heap keyed ``(time, priority, insertion-id)``; ``URGENT = 0`` <
  ``NORMAL = 1``;
* ``Timeout`` -> NORMAL at ``now + delay``; ``Process`` creation schedules an
  ``Initialize`` URGENT at ``now``; process termination schedules the process
  event NORMAL at ``now``; ``Event.succeed`` -> NORMAL at ``now``;
* ``run(until=t)`` schedules a stop event URGENT at ``t`` (so events at exactly
  ``t`` are NOT processed) and raises ``ValueError`` when ``t <= now``;
* ``step()`` pops ONE event and runs its callbacks in registration order;
* a process that yields an already *processed* event continues immediately, a
  triggered-but-unprocessed one waits for the pop;
* ``Store``/``Container`` are ``BaseResource`` queues: put/get events are
  appended to FIFO queues and served head-first with head-of-line blocking;
  every processed put re-triggers the get queue and vice versa.

Only the API surface the reference and its tests touch is provided
(SURVEY.md section 8c): ``Environment(now, active_process, process, timeout,
event, schedule, peek, step, run)``, ``Event``, ``Timeout``, ``Process``,
``Store(put, get, items, capacity)``, ``Container(put, get, level, capacity)``,
``AllOf/AnyOf``, ``Interrupt``.

Nothing in the product path (``asyncflow_amd``) imports this module.
"""

from __future__ import annotations

from heapq import heappop, heappush
from itertools import count
from types import GeneratorType
from typing import Any, Callable, Generator, Iterable, List, Optional

__version__ = "4.1.1+standin"

Infinity = float("inf")

URGENT = 0
NORMAL = 1


class _Pending:
    def __repr__(self) -> str:  # pragma: no cover - cosmetic
        return "<PENDING>"


PENDING = _Pending()


# --------------------------------------------------------------------------- #
# Exceptions                                                                   #
# --------------------------------------------------------------------------- #
class SimPyException(Exception):
    """Base class of all SimPy specific exceptions."""


class Interrupt(SimPyException):
    """Thrown into a process when it is interrupted."""

    @property
    def cause(self) -> Any:
        return self.args[0]


class StopSimulation(Exception):
    """Raised by the stop event's callback to leave ``Environment.run``."""

    @classmethod
    def callback(cls, event: "Event") -> None:
        if event.ok:
            raise cls(event.value)
        raise event._value  # noqa: SLF001


class EmptySchedule(Exception):
    """Raised by ``step`` when no events are left."""


# --------------------------------------------------------------------------- #
# Events                                                                       #
# --------------------------------------------------------------------------- #
class Event:
    """An event that may happen at some point in time."""

    def __init__(self, env: "Environment") -> None:
        self.env = env
        self.callbacks: Optional[List[Callable[["Event"], None]]] = []
        self._value: Any = PENDING
        self._ok: bool = True
        self._defused: bool = False

    @property
    def triggered(self) -> bool:
        return self._value is not PENDING

    @property
    def processed(self) -> bool:
        return self.callbacks is None

    @property
    def ok(self) -> bool:
        return self._ok

    @property
    def defused(self) -> bool:
        return self._defused

    @defused.setter
    def defused(self, value: bool) -> None:
        self._defused = bool(value)

    @property
    def value(self) -> Any:
        if self._value is PENDING:
            msg = f"Value of {self} is not yet available"
            raise AttributeError(msg)
        return self._value

    def trigger(self, event: "Event") -> None:
        self._ok = event._ok
        self._value = event._value
        self.env.schedule(self)

    def succeed(self, value: Any = None) -> "Event":
        if self._value is not PENDING:
            msg = f"{self} has already been triggered"
            raise RuntimeError(msg)
        self._ok = True
        self._value = value
        self.env.schedule(self)
        return self

    def fail(self, exception: BaseException) -> "Event":
        if self._value is not PENDING:
            msg = f"{self} has already been triggered"
            raise RuntimeError(msg)
        if not isinstance(exception, BaseException):
            msg = f"{exception} is not an exception."
            raise TypeError(msg)
        self._ok = False
        self._value = exception
        self.env.schedule(self)
        return self

    def __and__(self, other: "Event") -> "Condition":
        return Condition(self.env, Condition.all_events, [self, other])

    def __or__(self, other: "Event") -> "Condition":
        return Condition(self.env, Condition.any_events, [self, other])


class Timeout(Event):
    """Event that gets processed after *delay* has passed (NORMAL priority)."""

    def __init__(self, env: "Environment", delay: float, value: Any = None) -> None:
        if delay < 0:
            msg = f"Negative delay {delay}"
            raise ValueError(msg)
        self.env = env
        self.callbacks = []
        self._value = value
        self._delay = delay
        self._ok = True
        self._defused = False
        env.schedule(self, NORMAL, delay)


class Initialize(Event):
    """Starts a process: URGENT at the current time."""

    def __init__(self, env: "Environment", process: "Process") -> None:
        self.env = env
        self.callbacks = [process._resume]  # noqa: SLF001
        self._value = None
        self._ok = True
        self._defused = False
        env.schedule(self, URGENT)


class Interruption(Event):
    """Immediately schedules an Interrupt to be thrown into *process*."""

    def __init__(self, process: "Process", cause: Any) -> None:
        self.env = process.env
        self.callbacks = [self._interrupt]
        self._value = Interrupt(cause)
        self._ok = False
        self._defused = True
        if process._value is not PENDING:  # noqa: SLF001
            msg = f"{process} has terminated and cannot be interrupted."
            raise RuntimeError(msg)
        if process is self.env.active_process:
            msg = "A process is not allowed to interrupt itself."
            raise RuntimeError(msg)
        self.process = process
        self.env.schedule(self, URGENT)

    def _interrupt(self, event: Event) -> None:
        if self.process._value is not PENDING:  # noqa: SLF001
            return
        target = self.process._target  # noqa: SLF001
        if target is not None and target.callbacks is not None:
            try:
                target.callbacks.remove(self.process._resume)  # noqa: SLF001
            except ValueError:
                pass
        self.process._resume(self)  # noqa: SLF001


class Process(Event):
    """Process an event-yielding generator."""

    def __init__(self, env: "Environment", generator: Generator) -> None:
        if not hasattr(generator, "throw"):
            msg = f"{generator} is not a generator."
            raise ValueError(msg)
        self.env = env
        self.callbacks = []
        self._value = PENDING
        self._ok = True
        self._defused = False
        self._generator = generator
        self._target: Optional[Event] = Initialize(env, self)

    @property
    def target(self) -> Optional[Event]:
        return self._target

    @property
    def name(self) -> str:
        return getattr(self._generator, "__name__", "process")

    @property
    def is_alive(self) -> bool:
        return self._value is PENDING

    def interrupt(self, cause: Any = None) -> None:
        Interruption(self, cause)

    def _resume(self, event: Event) -> None:
        env = self.env
        env._active_proc = self  # noqa: SLF001
        while True:
            try:
                if event._ok:  # noqa: SLF001
                    event = self._generator.send(event._value)  # noqa: SLF001
                else:
                    event._defused = True  # noqa: SLF001
                    exc = type(event._value)(*event._value.args)  # noqa: SLF001
                    exc.__cause__ = event._value  # noqa: SLF001
                    event = self._generator.throw(exc)
            except StopIteration as stop:
                event = None  # type: ignore[assignment]
                self._ok = True
                self._value = stop.args[0] if len(stop.args) else None
                env.schedule(self)
                break
            except BaseException as exc:  # noqa: BLE001
                event = None  # type: ignore[assignment]
                self._ok = False
                self._value = exc
                env.schedule(self)
                break

            try:
                if event.callbacks is not None:
                    # not yet processed: wait for it
                    event.callbacks.append(self._resume)
                    break
            except AttributeError:
                if not hasattr(event, "callbacks"):
                    msg = f'Invalid yield value "{event}"'
                    descr = RuntimeError(msg)
                    event = Event(env)
                    event._ok = False  # noqa: SLF001
                    event._value = descr  # noqa: SLF001
                    event._defused = False  # noqa: SLF001
                    continue
                raise
            # already processed: continue immediately with its value

        self._target = event
        env._active_proc = None  # noqa: SLF001


class ConditionValue:
    """Result of a Condition: ordered mapping event -> value."""

    def __init__(self) -> None:
        self.events: List[Event] = []

    def __getitem__(self, key: Event) -> Any:
        if key not in self.events:
            raise KeyError(str(key))
        return key._value  # noqa: SLF001

    def __contains__(self, key: Event) -> bool:
        return key in self.events

    def __eq__(self, other: object) -> bool:
        if isinstance(other, ConditionValue):
            return self.events == other.events
        return self.todict() == other

    def __iter__(self):
        return iter(self.events)

    def keys(self):
        return (e for e in self.events)

    def values(self):
        return (e._value for e in self.events)  # noqa: SLF001

    def items(self):
        return ((e, e._value) for e in self.events)  # noqa: SLF001

    def todict(self) -> dict:
        return {e: e._value for e in self.events}  # noqa: SLF001


class Condition(Event):
    """Event triggered once *evaluate(events, count)* is true."""

    def __init__(self, env: "Environment", evaluate: Callable, events: Iterable[Event]) -> None:
        super().__init__(env)
        self._evaluate = evaluate
        self._events = tuple(events)
        self._count = 0
        if not self._events:
            self.succeed(ConditionValue())
            return
        for ev in self._events:
            if self.env != ev.env:
                msg = "It is not allowed to mix events from different environments"
                raise ValueError(msg)
        for ev in self._events:
            if ev.callbacks is None:
                self._check(ev)
            else:
                ev.callbacks.append(self._check)
        assert isinstance(self.callbacks, list)
        self.callbacks.append(self._build_value)

    def _populate_value(self, value: ConditionValue) -> None:
        for ev in self._events:
            if isinstance(ev, Condition):
                ev._populate_value(value)  # noqa: SLF001
            elif ev.callbacks is None:
                value.events.append(ev)

    def _build_value(self, event: Event) -> None:
        self._remove_check_callbacks()
        if event._ok:  # noqa: SLF001
            self._value = ConditionValue()
            self._populate_value(self._value)

    def _remove_check_callbacks(self) -> None:
        for ev in self._events:
            if ev.callbacks and self._check in ev.callbacks:
                ev.callbacks.remove(self._check)
            if isinstance(ev, Condition):
                ev._remove_check_callbacks()  # noqa: SLF001

    def _check(self, event: Event) -> None:
        if self._value is not PENDING:
            return
        self._count += 1
        if not event._ok:  # noqa: SLF001
            event._defused = True  # noqa: SLF001
            self.fail(event._value)  # noqa: SLF001
        elif self._evaluate(self._events, self._count):
            self.succeed()

    @staticmethod
    def all_events(events: tuple, count_: int) -> bool:
        return len(events) == count_

    @staticmethod
    def any_events(events: tuple, count_: int) -> bool:
        return count_ > 0 or len(events) == 0


class AllOf(Condition):
    def __init__(self, env: "Environment", events: Iterable[Event]) -> None:
        super().__init__(env, Condition.all_events, events)


class AnyOf(Condition):
    def __init__(self, env: "Environment", events: Iterable[Event]) -> None:
        super().__init__(env, Condition.any_events, events)


# --------------------------------------------------------------------------- #
# Environment                                                                  #
# --------------------------------------------------------------------------- #
class Environment:
    """Execution environment: event heap + simulation clock."""

    def __init__(self, initial_time: float = 0) -> None:
        self._now = initial_time
        self._queue: list = []
        self._eid = count()
        self._active_proc: Optional[Process] = None

    @property
    def now(self) -> float:
        return self._now

    @property
    def active_process(self) -> Optional[Process]:
        return self._active_proc

    # factories ------------------------------------------------------------ #
    def process(self, generator: Generator) -> Process:
        return Process(self, generator)

    def timeout(self, delay: float = 0, value: Any = None) -> Timeout:
        return Timeout(self, delay, value)

    def event(self) -> Event:
        return Event(self)

    def all_of(self, events: Iterable[Event]) -> AllOf:
        return AllOf(self, events)

    def any_of(self, events: Iterable[Event]) -> AnyOf:
        return AnyOf(self, events)

    # scheduling ------------------------------------------------------------ #
    def schedule(self, event: Event, priority: int = NORMAL, delay: float = 0) -> None:
        heappush(self._queue, (self._now + delay, priority, next(self._eid), event))

    def peek(self) -> float:
        try:
            return self._queue[0][0]
        except IndexError:
            return Infinity

    def step(self) -> None:
        try:
            self._now, _, _, event = heappop(self._queue)
        except IndexError:
            raise EmptySchedule from None

        callbacks, event.callbacks = event.callbacks, None
        for callback in callbacks:
            callback(event)

        if not event._ok and not event._defused:  # noqa: SLF001
            exc = type(event._value)(*event._value.args)  # noqa: SLF001
            exc.__cause__ = event._value  # noqa: SLF001
            raise exc

    def run(self, until: Any = None) -> Any:
        if until is not None:
            if not isinstance(until, Event):
                at = until if isinstance(until, int) else float(until)
                if at <= self.now:
                    msg = f"until ({at}) must be greater than the current simulation time"
                    raise ValueError(msg)
                until = Event(self)
                until._ok = True  # noqa: SLF001
                until._value = None  # noqa: SLF001
                self.schedule(until, URGENT, at - self.now)
            elif until.callbacks is None:
                return until.value
            until.callbacks.append(StopSimulation.callback)

        try:
            while True:
                self.step()
        except StopSimulation as exc:
            return exc.args[0]
        except EmptySchedule:
            if until is not None:
                assert not until.triggered
                msg = (
                    f'No scheduled events left but "until" event was not triggered: {until}'
                )
                raise RuntimeError(msg) from None
        return None


# --------------------------------------------------------------------------- #
# Shared resources: BaseResource, Store, Container                             #
# --------------------------------------------------------------------------- #
class Put(Event):
    """Generic event for requesting to put something into a resource."""

    def __init__(self, resource: "BaseResource") -> None:
        super().__init__(resource._env)  # noqa: SLF001
        self.resource = resource
        self.proc = self.env.active_process
        resource.put_queue.append(self)
        self.callbacks.append(resource._trigger_get)  # noqa: SLF001
        resource._trigger_put(None)  # noqa: SLF001

    def __enter__(self) -> "Put":
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> Optional[bool]:
        self.cancel()
        return None

    def cancel(self) -> None:
        if not self.triggered:
            self.resource.put_queue.remove(self)


class Get(Event):
    """Generic event for requesting to get something from a resource."""

    def __init__(self, resource: "BaseResource") -> None:
        super().__init__(resource._env)  # noqa: SLF001
        self.resource = resource
        self.proc = self.env.active_process
        resource.get_queue.append(self)
        self.callbacks.append(resource._trigger_put)  # noqa: SLF001
        resource._trigger_get(None)  # noqa: SLF001

    def __enter__(self) -> "Get":
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> Optional[bool]:
        self.cancel()
        return None

    def cancel(self) -> None:
        if not self.triggered:
            self.resource.get_queue.remove(self)


class BaseResource:
    """FIFO put/get queues with head-of-line blocking."""

    def __init__(self, env: Environment, capacity: float) -> None:
        self._env = env
        self._capacity = capacity
        self.put_queue: List[Put] = []
        self.get_queue: List[Get] = []

    @property
    def capacity(self) -> float:
        return self._capacity

    def _do_put(self, event: Put) -> Optional[bool]:
        raise NotImplementedError

    def _do_get(self, event: Get) -> Optional[bool]:
        raise NotImplementedError

    def _trigger_put(self, get_event: Optional[Get]) -> None:
        idx = 0
        while idx < len(self.put_queue):
            put_event = self.put_queue[idx]
            proceed = self._do_put(put_event)
            if not put_event.triggered:
                idx += 1
            elif self.put_queue.pop(idx) != put_event:
                msg = "Put queue invariant violated"
                raise RuntimeError(msg)
            if not proceed:
                break

    def _trigger_get(self, put_event: Optional[Put]) -> None:
        idx = 0
        while idx < len(self.get_queue):
            get_event = self.get_queue[idx]
            proceed = self._do_get(get_event)
            if not get_event.triggered:
                idx += 1
            elif self.get_queue.pop(idx) != get_event:
                msg = "Get queue invariant violated"
                raise RuntimeError(msg)
            if not proceed:
                break


class StorePut(Put):
    def __init__(self, store: "Store", item: Any) -> None:
        self.item = item
        super().__init__(store)


class StoreGet(Get):
    pass


class Store(BaseResource):
    """FIFO store of Python objects with optional capacity."""

    def __init__(self, env: Environment, capacity: float = Infinity) -> None:
        if capacity <= 0:
            msg = '"capacity" must be > 0.'
            raise ValueError(msg)
        super().__init__(env, capacity)
        self.items: List[Any] = []

    def put(self, item: Any) -> StorePut:
        return StorePut(self, item)

    def get(self) -> StoreGet:
        return StoreGet(self)

    def _do_put(self, event: StorePut) -> Optional[bool]:  # type: ignore[override]
        if len(self.items) < self._capacity:
            self.items.append(event.item)
            event.succeed()
        return None

    def _do_get(self, event: StoreGet) -> Optional[bool]:  # type: ignore[override]
        if self.items:
            event.succeed(self.items.pop(0))
        return None


class ContainerPut(Put):
    def __init__(self, container: "Container", amount: float) -> None:
        if amount <= 0:
            msg = f"amount(={amount}) must be > 0."
            raise ValueError(msg)
        self.amount = amount
        super().__init__(container)


class ContainerGet(Get):
    def __init__(self, container: "Container", amount: float) -> None:
        if amount <= 0:
            msg = f"amount(={amount}) must be > 0."
            raise ValueError(msg)
        self.amount = amount
        super().__init__(container)


class Container(BaseResource):
    """Continuous/discrete level resource."""

    def __init__(self, env: Environment, capacity: float = Infinity, init: float = 0) -> None:
        if capacity <= 0:
            msg = '"capacity" must be > 0.'
            raise ValueError(msg)
        if init < 0:
            msg = '"init" must be >= 0.'
            raise ValueError(msg)
        if init > capacity:
            msg = '"init" must be <= "capacity".'
            raise ValueError(msg)
        super().__init__(env, capacity)
        self._level = init

    @property
    def level(self) -> float:
        return self._level

    def put(self, amount: float) -> ContainerPut:
        return ContainerPut(self, amount)

    def get(self, amount: float) -> ContainerGet:
        return ContainerGet(self, amount)

    def _do_put(self, event: ContainerPut) -> Optional[bool]:  # type: ignore[override]
        if self._capacity - self._level >= event.amount:
            self._level += event.amount
            event.succeed()
            return True
        return None

    def _do_get(self, event: ContainerGet) -> Optional[bool]:  # type: ignore[override]
        if self._level >= event.amount:
            self._level -= event.amount
            event.succeed()
            return True
        return None


__all__ = [
    "AllOf",
    "AnyOf",
    "Container",
    "Environment",
    "Event",
    "Infinity",
    "Interrupt",
    "Process",
    "Store",
    "Timeout",
]
