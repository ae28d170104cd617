"""Simulation driver: iterate a scheme, keep timers, run post-processes.

Host control flow only; same constructor and behaviour as the reference's
``triflow/core/simulation.py:160-438``:

* ``Simulation(model, fields, parameters, dt, t=0, tmax=None, id=None,
  hook=null_hook, scheme=schemes.RODASPR, time_stepping=True, **kwargs)``;
  ``kwargs`` are forwarded to the scheme constructor / the step-doubling
  wrapper according to their signatures (``simulation.py:165-197``);
* with ``time_stepping=True`` (the default) *every* scheme instance is wrapped
  in ``schemes.time_stepping``: the reference's guard compares the instance
  with a list of classes and is therefore always true
  (``simulation.py:190-197``).  Pass ``time_stepping=False`` for a plain
  fixed-step run;
* ``dt`` is clipped so that the last step lands on ``tmax``
  (``simulation.py:215-217``); iteration stops on ``isclose(t, tmax)``;
* a ``RuntimeError`` raised by the scheme sets ``status = 'failed'`` and is
  re-raised (``simulation.py:259-261``).

The displays of the reference are outside the hot path; ``stream`` is a minimal
in-process publisher (user callbacks, the persistence container of
``container.py``) for the emitted simulation states.
"""

import datetime
import inspect
import logging
import pprint
import time
import warnings
from collections import namedtuple
from uuid import uuid1

from numpy import isclose

from . import schemes
from .device import DirichletHook, null_hook

log = logging.getLogger(__name__)
log.addHandler(logging.NullHandler())

PostProcess = namedtuple("PostProcess", ["name", "function", "description"])


class Stream:
    """Tiny stand-in for ``streamz.Stream``: ``sink(callback)`` + ``emit(x)``."""

    def __init__(self):
        self._sinks = []

    def sink(self, callback):
        self._sinks.append(callback)
        return self

    def emit(self, value):
        for callback in self._sinks:
            callback(value)


class Timer:
    def __init__(self, last, total):
        self.last, self.total = last, total

    def __repr__(self):
        return "last:   %s\ntotal:  %s" % (datetime.timedelta(seconds=self.last),
                                           datetime.timedelta(seconds=self.total))


def _accepted_kwargs(kwargs, function):
    """The entries of ``kwargs`` that ``function`` names (simulation.py:165-174)."""
    names = inspect.signature(function).parameters
    return {k: v for k, v in kwargs.items() if k in names}


class Simulation:
    def __init__(self, model, fields, parameters, dt, t=0, tmax=None, id=None,
                 hook=null_hook, scheme=schemes.RODASPR, time_stepping=True, **kwargs):
        kwargs["time_stepping"] = time_stepping
        self.id = str(uuid1())[:6] if not id else id
        self.model = model
        self.parameters = parameters
        self.fields = model.fields_template(**{k: fields[k] for k in fields.keys()}) \
            if not isinstance(fields, dict) else model.fields_template(**fields)
        self.t = t
        self.user_dt = self.dt = dt
        self.tmax = tmax
        self.i = 0
        self._stream = Stream()
        self._pprocesses = []
        self._scheme = scheme(model, **_accepted_kwargs(kwargs, scheme.__init__))
        if time_stepping:
            self._scheme = schemes.time_stepping(
                self._scheme, **_accepted_kwargs(kwargs, schemes.time_stepping))
        self.status = "created"
        self._total_running = 0
        self._last_running = 0
        self._created_timestamp = datetime.datetime.now()
        self._started_timestamp = None
        self._last_timestamp = None
        self._actual_timestamp = datetime.datetime.now()
        self._hook = hook
        self._container = None
        self._iterator = self.compute()

    def _compute_one_step(self, t, fields, pars):
        if isinstance(self._hook, DirichletHook):
            # the reference calls hook here and again as the first action of the scheme
            # (simulation.py:214, schemes.py:145,549); a declarative hook is idempotent, so
            # only its parameter update is taken here and the fields stay on the GPU
            pars = self._hook.update_pars(t, pars)
        else:
            fields, pars = self._hook(t, fields, pars)
        self.dt = (self.tmax - t if self.tmax and (t + self.dt >= self.tmax) else self.dt)
        before = time.perf_counter()
        t, fields = self._scheme(t, fields, self.dt, pars, hook=self._hook)
        after = time.perf_counter()
        self._last_running = after - before
        self._total_running += self._last_running
        self._last_timestamp = self._actual_timestamp
        self._actual_timestamp = datetime.datetime.now()
        return t, fields, pars

    def compute(self):
        """Generator yielding ``(t, fields)`` after every ``dt``."""
        fields, t, pars = self.fields, self.t, self.parameters
        self._started_timestamp = datetime.datetime.now()
        self.stream.emit(self)
        try:
            while True:
                t, fields, pars = self._compute_one_step(t, fields, pars)
                self.i += 1
                self.t, self.fields, self.parameters = t, fields, pars
                for pprocess in self.post_processes:
                    pprocess.function(self)
                self.stream.emit(self)
                yield self.t, self.fields
                if self.tmax and isclose(self.t, self.tmax):
                    self._end_simulation()
                    return
        except RuntimeError:
            self.status = "failed"
            raise

    def _end_simulation(self):
        self.status = "finished"
        if self._container is not None:
            self._container.flush()
            self._container.merge()

    def run(self, progress=True, verbose=False):
        """Compute all steps (infinite without ``tmax``); returns the last ``(t, fields)``."""
        emit = log.info if verbose else log.debug
        total = int(self.tmax // self.user_dt) if self.tmax else None
        iterator = self
        if progress:
            try:
                from tqdm import tqdm
                iterator = tqdm(self, initial=min(self.i, total) if total else self.i,
                                total=total)
            except ImportError:
                pass
        last = None
        for t, fields in iterator:
            last = (t, fields)
            emit("%s running: t: %g" % (self.id, t))
        if last is None:
            warnings.warn("Simulation already ended")
        return last

    def __repr__(self):
        try:
            hook_source = inspect.getsource(self._hook)
        except (OSError, TypeError):
            hook_source = repr(self._hook)
        pars = "\n\t".join(("%s:" % k).ljust(12) + pprint.pformat(v)
                           for k, v in self.parameters.items())
        return ("{name:=^30}\n\ncreated:      {created}\nstarted:      {started}\n"
                "last:         {last}\n\ntime:         {t:g}\niteration:    {i:g}\n\n"
                "last step:    {step}\ntotal time:   {total}\n\n\n"
                "Physical parameters\n-------------------\n{pars}\n\n"
                "Hook function\n-------------\n{hook}\n\n"
                "=========== Model ===========\n{model}").format(
            name=" %s " % self.id, created=self._created_timestamp,
            started=self._started_timestamp, last=self._last_timestamp, t=self.t, i=self.i,
            step=datetime.timedelta(seconds=self._last_running) if self._last_running else None,
            total=datetime.timedelta(seconds=self._total_running), pars=pars,
            hook=hook_source, model=self.model)

    def attach_container(self, path=None, save="all", mode="w", nbuffer=50, force=False):
        """Keep the emitted states (reference simulation.py:352-381): in memory, and under
        ``path/<id>`` when a path is given.  See ``triflow_amd/container.py``."""
        from .container import TriflowContainer
        self._container = TriflowContainer("%s/%s" % (path, self.id) if path else None,
                                           save=save, mode=mode, metadata=self.parameters,
                                           force=force, nbuffer=nbuffer)
        self._container.connect(self.stream)
        return self._container

    @property
    def post_processes(self):
        return self._pprocesses

    @property
    def stream(self):
        return self._stream

    @property
    def container(self):
        return self._container

    @property
    def timer(self):
        return Timer(self._last_running, self._total_running)

    def add_post_process(self, name, post_process, description=""):
        """Register ``post_process(simulation)`` and run it once (simulation.py:399-423)."""
        self._pprocesses.append(PostProcess(name=name, function=post_process,
                                            description=description))
        self._pprocesses[-1].function(self)

    def remove_post_process(self, name):
        self._pprocesses = [p for p in self._pprocesses if p.name != name]

    def __iter__(self):
        return self.compute()

    def __next__(self):
        return next(self._iterator)
