"""Lower the SymPy stencil IR of a Model to the per-node C bodies that the
hand-written kernel skeleton (csrc/tf_kernels.h) is specialised with.

The reference hands ``model.F_array`` / ``model._J_sparse_array`` to
``sympy.lambdify`` and lets NumPy evaluate the printed expression
(``triflow/core/compilers.py:207-219``).  To compute *the same floating-point
values*, this lowering starts from the very string SymPy's NumPy printer
produces for that call, parses it with ``ast`` and emits C with the identical
operation tree (no re-association, no common-subexpression rewriting, no FMA
contraction; the C compiler may only share bit-identical subtrees).  Name
overrides follow the reference's module dictionary: ``Heaviside`` is
identically one (``compilers.py:204-205``), ``Max``/``Min`` are
``numpy.maximum``/``minimum`` chains.
"""

import ast
import hashlib
import inspect

import numpy as np
from sympy import lambdify

_FUNCS_1 = {"sqrt": "sqrt", "exp": "exp", "log": "log", "sin": "sin", "cos": "cos",
            "tan": "tan", "tanh": "tanh", "sinh": "sinh", "cosh": "cosh",
            "arctan": "atan", "arcsin": "asin", "arccos": "acos",
            "abs": "tf_abs", "absolute": "tf_abs", "fabs": "tf_abs", "sign": "tf_sign",
            "log10": "log10", "log2": "log2", "cbrt": "cbrt", "expm1": "expm1",
            "log1p": "log1p", "floor": "floor", "ceil": "ceil"}


class UnsupportedExpression(NotImplementedError):
    pass


def _dbl(value):
    """C literal with the exact binary value of a Python number."""
    f = float(value)
    if f != value and not isinstance(value, float):
        raise UnsupportedExpression("integer constant %r is not a double" % (value,))
    if f == int(f) and abs(f) < 2 ** 53:
        return "%d.0" % int(f)
    return f.hex()          # C99 / C++17 hexadecimal floating literal


class _CEmitter(ast.NodeVisitor):
    """Python expression AST -> C expression string.  Also tracks which
    sub-expressions are *uniform* (same value for every node a thread visits:
    dx, scalar parameters, constants) so that divisions by a uniform divisor can
    use the exact reciprocal form ``tf_div_u`` with the divisor and its
    reciprocal hoisted out of the node loop."""

    def __init__(self, names, uniform_names=()):
        self.names = names          # python identifier -> C expression
        self.uniform_names = set(uniform_names)
        self.denominators = {}      # C expression -> index
        self.den_count = {}         # C expression -> number of quotients using it
        self.shared = {}            # den_count of a previous pass over the same expressions
        self.host_consts = {}       # python source of a uniform pow()/libm call -> index

    def host_const(self, node):
        """A node-independent power or libm call (``dx**3``, ``exp(k)``): evaluated on
        the host with NumPy -- i.e. by the very libm the reference's lambdified code
        calls -- and handed to the kernel as an extra scalar, so its bits match."""
        src = ast.unparse(node)
        k = self.host_consts.setdefault(src, len(self.host_consts))
        return "tf_hc[%d]" % k

    def is_uniform(self, node):
        if isinstance(node, ast.Constant):
            return True
        if isinstance(node, ast.Name):
            return node.id in self.uniform_names or node.id in ("pi", "E", "e")
        if isinstance(node, ast.UnaryOp):
            return self.is_uniform(node.operand)
        if isinstance(node, ast.BinOp):
            return self.is_uniform(node.left) and self.is_uniform(node.right)
        if isinstance(node, ast.Call):
            args = []
            for a in node.args:
                args.extend(a.elts if isinstance(a, (ast.List, ast.Tuple)) else [a])
            return all(self.is_uniform(a) for a in args
                       if not isinstance(a, (ast.Name, ast.Attribute)) or
                       not (getattr(a, "id", getattr(a, "attr", "")) in ("maximum", "minimum")))
        return False

    def visit_Name(self, node):
        if node.id in self.names:
            return self.names[node.id]
        if node.id == "pi":
            return _dbl(np.pi)
        if node.id in ("E", "e"):           # SymPy's Euler number, printed as numpy.e
            return _dbl(np.e)
        raise UnsupportedExpression("unknown symbol %r in stencil expression" % node.id)

    def visit_Constant(self, node):
        if isinstance(node.value, bool) or not isinstance(node.value, (int, float)):
            raise UnsupportedExpression("constant %r" % (node.value,))
        return _dbl(node.value)

    def visit_UnaryOp(self, node):
        operand = self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            return "(-%s)" % operand
        if isinstance(node.op, ast.UAdd):
            return operand
        raise UnsupportedExpression(ast.dump(node))

    def visit_BinOp(self, node):
        if isinstance(node.op, ast.Pow):
            return self._pow(node)
        if isinstance(node.op, ast.Div) and not self.is_uniform(node.left):
            den = self.visit(node.right)
            self.den_count[den] = self.den_count.get(den, 0) + 1
            # uniform divisor: reciprocal hoisted out of the node loop; a
            # node-dependent divisor shared by several quotients of the same node:
            # one true division for the reciprocal, then the exact 3-op quotient
            if self.is_uniform(node.right) or self.shared.get(den, 0) >= 2:
                k = self.denominators.setdefault(den, len(self.denominators))
                return "tf_div_u(%s, tf_den%d, tf_rden%d)" % (self.visit(node.left), k, k)
        ops = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}
        for klass, sym in ops.items():
            if isinstance(node.op, klass):
                return "(%s %s %s)" % (self.visit(node.left), sym, self.visit(node.right))
        raise UnsupportedExpression(ast.dump(node))

    def _pow(self, node):
        if self.is_uniform(node) and not isinstance(node.left, ast.Constant):
            return self.host_const(node)
        base = self.visit(node.left)
        expo = node.right
        neg = False
        if isinstance(expo, ast.UnaryOp) and isinstance(expo.op, ast.USub):
            neg, expo = True, expo.operand
        if isinstance(expo, ast.Constant) and isinstance(expo.value, (int, float)) \
                and not isinstance(expo.value, bool):
            val = -expo.value if neg else expo.value
            if val == 2:
                return "tf_sq(%s)" % base          # numpy.power fast path: square
            if val == 0.5:
                return "sqrt(%s)" % base           # numpy.power fast path: sqrt
            if val == 1:
                return base
            if val == -1:
                return "(1.0 / %s)" % base         # numpy.power fast path: reciprocal
            if float(val) == int(val) and abs(val) <= 16:
                return "tf_powi(%s, %d)" % (base, int(val))
            return "pow(%s, %s)" % (base, _dbl(val))
        return "pow(%s, %s)" % (base, self.visit(node.right))

    def visit_Call(self, node):
        name = node.func.attr if isinstance(node.func, ast.Attribute) else node.func.id
        if name == "Heaviside":
            return "1.0"                           # reference compilers.py:204-205
        if name == "reduce":                       # reduce(maximum, [a, b, ...])
            op = node.args[0]
            opname = op.attr if isinstance(op, ast.Attribute) else op.id
            fn = {"maximum": "tf_max", "minimum": "tf_min"}.get(opname)
            if fn is None or not isinstance(node.args[1], (ast.List, ast.Tuple)):
                raise UnsupportedExpression("reduce(%s, ...)" % opname)
            items = [self.visit(e) for e in node.args[1].elts]
            out = items[0]
            for item in items[1:]:
                out = "%s(%s, %s)" % (fn, out, item)
            return out
        if name in ("amax", "amin"):               # older SymPy: amax((a, b), axis=0)
            fn = "tf_max" if name == "amax" else "tf_min"
            items = [self.visit(e) for e in node.args[0].elts]
            out = items[0]
            for item in items[1:]:
                out = "%s(%s, %s)" % (fn, out, item)
            return out
        if name in ("maximum", "minimum"):
            fn = "tf_max" if name == "maximum" else "tf_min"
            return "%s(%s, %s)" % (fn, self.visit(node.args[0]), self.visit(node.args[1]))
        if name in _FUNCS_1 and len(node.args) == 1:
            if self.is_uniform(node.args[0]) and not isinstance(node.args[0], ast.Constant) \
                    and name not in ("abs", "absolute", "fabs", "sign", "floor", "ceil", "sqrt"):
                return self.host_const(node)
            return "%s(%s)" % (_FUNCS_1[name], self.visit(node.args[0]))
        raise UnsupportedExpression("function %r is not supported by the HIP compiler" % name)

    def generic_visit(self, node):
        raise UnsupportedExpression(ast.dump(node))


def _printed_expressions(symbolic_args, exprs):
    """The element expressions of the list SymPy's lambdify would evaluate."""
    if not exprs:
        return []
    func = lambdify(symbolic_args, exprs, modules=[{"Heaviside": lambda *a: 1}, "numpy"],
                    cse=False)
    src = inspect.getsource(func)
    tree = ast.parse(src)
    fdef = tree.body[0]
    ret = fdef.body[-1]
    if len(fdef.body) != 1 or not isinstance(ret, ast.Return) \
            or not isinstance(ret.value, (ast.List, ast.Tuple)):
        raise UnsupportedExpression("unexpected lambdify output:\n" + src)
    return list(ret.value.elts)


def _c_ident(name):
    return "v_" + "".join(ch if ch.isalnum() else "_" for ch in name)


def lower_model(model, parvec_mask=0, seg=8, sweep_block=64):
    """Returns ``(source, spec)``: the per-model translation unit (without the
    skeleton includes' contents) and the dict of constants the runtime needs."""
    nvar = model._nvar
    fields = list(model._dep_vars) + list(model._help_funcs)
    nh = len(model._help_funcs)
    pars = list(model._pars)
    lo, hi = model._bounds
    mp = (model._window_range - 1) // 2
    if -lo != hi or hi != mp:
        raise UnsupportedExpression("asymmetric stencil window %r" % (model._bounds,))
    if mp < 1:
        mp = 1          # purely local models still get the width-3 skeleton
    if nvar + nh > 16 or len(pars) > 16:
        raise UnsupportedExpression("too many fields / parameters for the HIP skeleton")
    if mp * nvar > 16:
        # (the reduced levels of the banded solver are written for blocks of up to 16 x 16: 8 or 16 lanes
        # share a block row, tf_coop_hip.h; the reference takes any size, triflow/core/model.py:138-150)
        raise UnsupportedExpression(
            "the banded solver of the HIP back end handles b = (stencil half width) x (number of dependent "
            "variables) <= 16; this model has b = %d x %d = %d" % (mp, nvar, mp * nvar))
    sparse = [int(k) for k in model._sparse_indices[0]]
    nnz = len(sparse)
    real_mp = (model._window_range - 1) // 2
    pat_eq = [k % nvar for k in sparse]
    pat_var = [(k // nvar) % nvar for k in sparse]
    pat_off = [(k // nvar) // nvar - real_mp for k in sparse]

    # python identifier (as printed by SymPy) -> C identifier
    names = {}
    decls = []
    for f, name in enumerate(fields):
        for off in range(-mp, mp + 1):
            key = name if off == 0 else "%s_%s%d" % (name, "m" if off < 0 else "p", abs(off))
            names[key] = _c_ident(key)
            decls.append("const double %s = w[%d][%d];" % (_c_ident(key), f, off + mp))
    for k, name in enumerate(pars):
        names[name] = _c_ident(name)
        decls.append("const double %s = par[%d];" % (_c_ident(name), k))
    names["dx"] = "dx"
    names["x"] = "xc"

    uniform = {"dx"} | {name for k, name in enumerate(pars) if not (parvec_mask >> k) & 1}
    f_nodes = _printed_expressions(model._symbolic_args, model.F_array.tolist())
    j_nodes = _printed_expressions(model._symbolic_args, model._J_sparse_array.tolist())
    host_consts = {}

    def emit_all(nodes):
        first = _CEmitter(names, uniform)           # pass 1: count divisor reuse
        for n in nodes:
            first.visit(n)
        second = _CEmitter(names, uniform)
        second.shared = first.den_count
        second.host_consts = host_consts            # one table for F and J
        return second, [second.visit(n) for n in nodes]

    emit_f, f_c = emit_all(f_nodes)
    emit_j, j_c = emit_all(j_nodes)
    uses_x = any("xc" in _tokens(s) for s in f_c + j_c)

    # Jacobian entries that are the same at every node of a system (constant coefficients:
    # only dx, scalar parameters and constants): the solver kernels evaluate them once per
    # thread instead of reading them back from the value table at every node
    j_uniform = [1 if emit_j.is_uniform(n) else 0 for n in j_nodes]

    j_alias, j_alias_scale = _proportional_entries(model, j_uniform)

    def body(outname, exprs, emitter, only=None):
        lines = ["    " + d for d in decls if only is None or d.split("=")[1].strip().startswith("par[")]
        lines += ["    const double* tf_hc = par + %d;" % len(pars)]
        lines += ["    (void)dx; (void)xc; (void)par; (void)tf_hc;"]
        # node-independent divisors and their reciprocals: loop invariant, hoisted
        # (uniform entries divide uniform numerators: plain division, no hoisted divisor)
        for den, k in sorted(emitter.denominators.items(), key=lambda kv: kv[1]) if only is None else ():
            lines += ["    const double tf_den%d = %s;" % (k, den),
                      "    const double tf_rden%d = 1.0 / tf_den%d;" % (k, k)]
        lines += ["    %s[%d] = %s;" % (outname, i, e) for i, e in enumerate(exprs)
                  if only is None or only[i]]
        return "\n".join(lines)

    def arr(name, values, ctype="int"):
        vals = ", ".join(str(v) for v in values) if values else "0"
        return "static constexpr %s %s[%d] = {%s};" % (ctype, name, max(len(values), 1), vals)

    hc_list = [src_ for src_, _ in sorted(host_consts.items(), key=lambda kv: kv[1])]
    par_is_vec = [1 if (parvec_mask >> k) & 1 else 0 for k in range(len(pars))] + [0] * len(hc_list)
    if len(pars) + len(hc_list) > 16:
        raise UnsupportedExpression("too many parameters + host constants for the HIP skeleton")
    src = "\n".join([
        "// generated by triflow_amd.codegen -- do not edit",
        "// equations: " + " ; ".join(str(e) for e in model._diff_eqs),
        "#define TF_NVAR %d" % nvar,
        "#define TF_NH %d" % nh,
        "#define TF_MP %d" % mp,
        "#define TF_NNZ %d" % nnz,
        "#define TF_NPAR %d" % (len(pars) + len(host_consts)),
        "#define TF_SEG %d" % seg,
        "#define TF_SWEEP_BLOCK %d" % sweep_block,
        "#define TF_USES_X %d" % (1 if uses_x else 0),
        arr("tf_pat_eq", pat_eq), arr("tf_pat_var", pat_var), arr("tf_pat_off", pat_off),
        arr("tf_par_is_vec", par_is_vec, "bool"),
        arr("tf_j_uniform", j_uniform, "bool"),
        arr("tf_j_alias", j_alias), arr("tf_j_alias_scale", [_dbl(v) for v in j_alias_scale], "double"),
        "TF_DEVICE void tf_eval_F(const double (&w)[TF_NVAR + TF_NH][2 * TF_MP + 1], "
        "const double* par, double dx, double xc, double* F) {",
        body("F", f_c, emit_f),
        "}",
        "TF_DEVICE void tf_eval_J(const double (&w)[TF_NVAR + TF_NH][2 * TF_MP + 1], "
        "const double* par, double dx, double xc, double* J) {",
        body("J", j_c, emit_j),
        "}",
        "// the node-independent entries only (tf_j_uniform); same expressions, same bits",
        "TF_DEVICE void tf_eval_J_uniform(const double* par, double dx, double* J) {",
        "    const double xc = 0.0;",
        body("J", j_c, emit_j, only=j_uniform),
        "}",
        ""])
    spec = dict(nvar=nvar, nh=nh, npar=len(pars) + len(hc_list), npar_model=len(pars),
                host_consts=hc_list, mp=mp, nnz=nnz, seg=seg,
                sweep_block=sweep_block, uses_x=int(uses_x), parvec_mask=int(parvec_mask),
                b2=mp * nvar, pat_eq=pat_eq, pat_var=pat_var, pat_off=pat_off, j_uniform=j_uniform,
                j_alias=j_alias, j_alias_scale=j_alias_scale,
                fields=fields, pars=pars)
    return src, spec


def _proportional_entries(model, j_uniform):
    """Jacobian entries that are an exact power-of-two multiple of an earlier node-dependent
    entry (``-q*dxT/h`` differentiated with respect to ``T_m1`` and ``T_p1``; ``We*h*dxxxh`` with
    respect to the four neighbours of ``h``; a reaction term that enters two equations with
    opposite signs): the kernels that read the value table back load one of them and scale.
    ``alias[k]`` = index of the entry to load (-1: none), ``scale[k]`` = +-2**n.  Accepted only if
    (i) SymPy proves the proportionality and (ii) the two expressions, evaluated the way the
    reference evaluates them (the lambdified NumPy code whose operation tree the C code repeats),
    agree *bit for bit* on random inputs -- scaling by a power of two commutes with every
    rounding, a different association of the same product would not."""
    exprs = model._J_sparse_array.tolist()
    nnz = len(exprs)
    alias, scale = [-1] * nnz, [1.0] * nnz
    cand = [k for k in range(nnz) if not j_uniform[k]]
    if len(cand) < 2:
        return alias, scale
    func = lambdify(model._symbolic_args, exprs, modules=[{"Heaviside": lambda *a: 1}, "numpy"], cse=False)
    rng = np.random.default_rng(12345)
    samples = []
    for _ in range(2):
        args = [rng.uniform(0.5, 2.0, 257) * rng.choice([-1.0, 1.0], 257) for _ in model._symbolic_args]
        with np.errstate(all="ignore"):
            vals = func(*args)
        samples.append([np.broadcast_to(np.asarray(v, dtype=float), (257,)) for v in vals])
    for k in cand:
        for m in cand:
            if m >= k:
                break
            if alias[m] >= 0:
                continue
            a0, b0 = samples[0][k], samples[0][m]
            if not (np.isfinite(a0).all() and np.isfinite(b0).all()) or not np.all(b0 != 0.0):
                continue
            r = a0[0] / b0[0]
            mant, _ = np.frexp(abs(r))
            if not (np.isfinite(r) and r != 0.0 and mant == 0.5):
                continue
            if not all(np.array_equal(smp[k], r * smp[m]) for smp in samples):
                continue
            if sympy_simplify(exprs[k] - sympy_Float_exact(r) * exprs[m]) != 0:
                continue
            alias[k], scale[k] = m, float(r)
            break
    return alias, scale


def sympy_Float_exact(r):
    import sympy
    return sympy.Rational(*float(r).as_integer_ratio())


def sympy_simplify(e):
    import sympy
    return sympy.simplify(e)


_HOST_NS = {name: getattr(np, name) for name in
            ("sqrt", "exp", "log", "sin", "cos", "tan", "tanh", "sinh", "cosh", "arctan", "arcsin",
             "arccos", "log10", "log2", "cbrt", "expm1", "log1p", "maximum", "minimum", "pi")}
_HOST_NS["E"] = _HOST_NS["e"] = np.e


def eval_host_constants(spec, dx, par_values):
    """Values of the model's uniform power / libm sub-expressions for one system,
    computed exactly as the reference's lambdified NumPy code computes them."""
    env = {name: np.float64(np.ravel(v)[0]) for name, v in zip(spec["pars"], par_values)}
    env["dx"] = np.float64(dx)
    out = []
    for src_ in spec["host_consts"]:
        with np.errstate(all="ignore"):
            out.append(float(eval(src_, {"__builtins__": {}, **_HOST_NS}, env)))
    return out


def _tokens(text):
    out, cur = set(), ""
    for ch in text:
        if ch.isalnum() or ch == "_":
            cur += ch
        else:
            if cur:
                out.add(cur)
            cur = ""
    if cur:
        out.add(cur)
    return out


def source_hash(*parts):
    h = hashlib.sha256()
    for part in parts:
        h.update(part if isinstance(part, bytes) else str(part).encode())
        h.update(b"\0")
    return h.hexdigest()[:20]
