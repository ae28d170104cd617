"""Resident device state behind the schemes (host side of seam #2).

A :class:`Stepper` owns the state slots of one ``tf_solver`` and decides, for
every ``scheme(t, fields, dt, pars, hook)`` call, whether the incoming
``fields`` are already on the GPU (they are the container a previous step of
this stepper returned and nobody touched them on the host) or have to be
uploaded.  Containers returned by a step are *device backed*
(``fields.py``): their arrays are downloaded only when read.

``hook`` handling (reference call sites ``schemes.py:139,145,549,558``):

* ``null_hook``                      nothing to do;
* a :class:`DirichletHook`           applied by a kernel at the places the
                                     reference calls ``hook`` (no host traffic);
* any other callable                 the generic, slow path: the fields are
                                     brought to the host, the callable runs
                                     there, the result is uploaded again.
"""

import weakref

import numpy as np


def null_hook(t, fields, pars):
    return fields, pars


class DirichletHook:
    """Declarative hook, e.g. ``DirichletHook(U={0: 1.0, -1: 0.0})``.

    Equivalent to the reference idiom (``README.md:126-129``)::

        def hook(t, fields, pars):
            fields["U"][0] = 1; fields["U"][-1] = 0
            return fields, pars

    and usable as such on host containers, but recognised by the device schemes
    and applied in place on the GPU, at the places and with the times the
    reference calls ``hook`` (``schemes.py:139,145,549,558``).  A value may be a
    callable of ``t`` (time-dependent boundary data); ``parameters`` may be a
    callable ``(t, pars) -> dict`` of parameter updates (time-dependent
    coefficients) -- neither needs the fields on the host.
    """

    def __init__(self, parameters=None, **values):
        self.values = {var: dict(nodes) for var, nodes in values.items()}
        self.parameters = parameters

    def update_pars(self, t, pars):
        if self.parameters is None:
            return pars
        new = dict(pars)
        new.update(self.parameters(t, pars))
        return new

    def __call__(self, t, fields, pars):
        for var, nodes in self.values.items():
            for node, value in nodes.items():
                fields[var][node] = value(t) if callable(value) else value
        return fields, self.update_pars(t, pars)

    def layout(self, dependent_variables):
        """Static part: (variable index, node) of every entry."""
        dep = list(dependent_variables)
        return tuple((dep.index(var), int(node))
                     for var, nodes in self.values.items() for node in nodes)

    def values_at(self, t):
        return [float(v(t)) if callable(v) else float(v)
                for nodes in self.values.values() for v in nodes.values()]

    @property
    def time_dependent(self):
        return any(callable(v) for nodes in self.values.values() for v in nodes.values())

    def entries(self, dependent_variables, t=0.0):
        return [(i, n, v) for (i, n), v in zip(self.layout(dependent_variables),
                                               self.values_at(t))]


class DeviceBacking:
    """What a device-backed Fields container points at."""

    def __init__(self, stepper, slot, version):
        self.stepper, self.slot, self.version = stepper, slot, version

    def valid(self):
        return self.stepper.slot_version[self.slot] == self.version

    def register(self, fields):
        """Another container (a copy) now depends on this slot."""
        self.stepper._users[self.slot].append(weakref.ref(fields))

    def download(self, fields):
        if not self.valid():
            raise RuntimeError("device state of this Fields container was recycled "
                               "(internal error: it should have been saved first)")
        self.stepper.download_into(fields, self.slot)


class Stepper:
    def __init__(self, compiled, N, periodic, parvec_mask, nstate=6, **opts):
        self.compiled = compiled
        self.solver = compiled.solver(N, periodic, 1, parvec_mask, nstate=nstate, **opts)
        self.nstate = nstate
        self.slot_version = [0] * nstate
        self._users = [[] for _ in range(nstate)]       # weakrefs of attached containers
        self._age = [0] * nstate
        self._clock = 0
        self._bound_pars = None
        self._bound_x = None
        self._dirichlet = None

    # ---- slots ------------------------------------------------------------------
    def _live_users(self, slot):
        alive = []
        for ref in self._users[slot]:
            f = ref()
            if f is not None:
                b = f._device_backing()
                if b is not None and b.stepper is self and b.slot == slot \
                        and b.version == self.slot_version[slot]:
                    alive.append(f)
        self._users[slot] = [weakref.ref(f) for f in alive]
        return alive

    def free_slot(self, exclude=()):
        """A slot nobody refers to; if every slot is referenced, the least
        recently written one is saved to its containers first."""
        candidates = [s for s in range(self.nstate) if s not in exclude]
        for s in sorted(candidates, key=lambda k: self._age[k]):
            if not self._live_users(s):
                return self._claim(s)
        s = min(candidates, key=lambda k: self._age[k])
        for f in self._live_users(s):
            f._materialise()
        return self._claim(s)

    def _claim(self, slot):
        self.slot_version[slot] += 1
        self._clock += 1
        self._age[slot] = self._clock
        self._users[slot] = []
        return slot

    def resident_slot(self, fields):
        b = _backing_of(fields)
        if b is not None and b.stepper is self and b.valid():
            return b.slot
        return None

    def acquire(self, fields, exclude=()):
        """Slot holding the dependent variables of ``fields`` (uploads if needed)."""
        slot = self.resident_slot(fields)
        if slot is not None:
            writes = fields._take_point_writes() if hasattr(fields, "_take_point_writes") else []
            if not writes:
                return slot
            # a Python hook assigned single nodes (fields.U[0] = 1): the container is a copy
            # of the state it came from, so the writes go to a slot of its own
            dep = list(self.compiled.model._dep_vars)
            dst = self.free_slot(exclude=(slot, *exclude))
            self.solver.copy_state(slot, dst)
            self.solver.poke(dst, [(dep.index(k), i, v) for k, i, v in writes])
            backing = DeviceBacking(self, dst, self.slot_version[dst])
            fields._attach_device(backing)
            self._users[dst].append(weakref.ref(fields))
            return dst
        slot = self.free_slot(exclude)
        dep = self.compiled.model._dep_vars
        self.solver.set_state(slot, np.array([np.asarray(fields[k]) for k in dep]))
        return slot

    def wrap(self, template, slot):
        """New container for the state in ``slot`` (coordinates and help functions
        shared with ``template`` until it is read)."""
        if not hasattr(template, "_device_child"):
            # a container of another package (the reference's xarray-based Fields):
            # its protocol is copy() + fill(uflat) (fields.py:122-183)
            new = template.copy()
            new.fill(self.solver.get_state_flat(slot)[0])
            return new
        new = template._device_child(DeviceBacking(self, slot, self.slot_version[slot]))
        self._users[slot].append(weakref.ref(new))
        return new

    def download_into(self, fields, slot):
        arr = self.solver.get_state(slot)
        fields._fill_from_device(self.compiled.model._dep_vars, arr[:, 0, :])

    # ---- inputs ------------------------------------------------------------------
    def bind(self, fields, pars):
        """dx / x / parameters / help functions -> device (skipped when unchanged)."""
        cm = self.compiled
        x = np.asarray(fields["x"])
        xkey = (x.size, float(x[0]), float(x[-1]))
        values = [pars[k] for k in cm.pars]        # KeyError for a missing parameter
        mask = self.solver.model.spec["parvec_mask"]
        scalar = all(np.ndim(v) == 0 or np.size(v) == 1 for v in values)
        pkey = tuple(float(np.ravel(v)[0]) for v in values) if scalar and not mask else None
        helper_free = cm.nh == 0
        if helper_free and pkey is not None and pkey == self._bound_pars and xkey == self._bound_x:
            return
        helpers = None if helper_free else [np.asarray(fields[k]) for k in cm.model._help_funcs]
        cm.bind_inputs(self.solver, x, values, helpers)
        self._bound_pars, self._bound_x = pkey, xkey

    def set_hook(self, hook, t=0.0, t_after=None):
        """Boundary values the step kernels apply before (``t``) and after
        (``t_after``) the step."""
        if not isinstance(hook, DirichletHook):
            if self._dirichlet:
                self.solver.set_dirichlet(())
                self._dirichlet = ()
            return
        layout = hook.layout(self.compiled.model._dep_vars)
        before = hook.values_at(t)
        after = hook.values_at(t if t_after is None else t_after)
        key = (layout, tuple(before), tuple(after))
        if key == self._dirichlet:
            return
        if self._dirichlet is None or not self._dirichlet or self._dirichlet[0] != layout:
            self.solver.set_dirichlet([(i, n, v) for (i, n), v in zip(layout, before)])
        if layout:
            self.solver.set_dirichlet_values(before, after)
        self._dirichlet = key


def _backing_of(fields):
    probe = getattr(fields, "_device_backing", None)
    return probe() if probe is not None else None


def stepper_for(model, fields, pars, **opts):
    """The Stepper matching ``fields`` / ``pars`` (cached on the compiled model)."""
    compiled = getattr(model, "_device", None)
    if compiled is None:
        # a Model class of another package compiled with ``compiler=hip_compiler``:
        # the plugin's F function carries the device side (routines.py:11 keeps it as _ufunc)
        compiled = getattr(getattr(getattr(model, "F", None), "_ufunc", None), "device_model", None)
    if compiled is None:
        raise RuntimeError(
            "this scheme runs on the GPU and needs a model compiled with the HIP "
            "compiler (Model(..., compiler='hip')); got a model without device code")
    b = _backing_of(fields)
    if b is not None and b.valid() and b.stepper.compiled is compiled:
        return b.stepper
    values = [pars[k] for k in compiled.pars]
    mask = compiled.parvec_mask_of(values)
    periodic = bool(pars["periodic"])
    key = (fields.size, periodic, mask, tuple(sorted(opts.items())))
    cache = compiled.__dict__.setdefault("_steppers", {})
    if key not in cache:
        cache[key] = Stepper(compiled, fields.size, periodic, mask, **opts)
    return cache[key]
