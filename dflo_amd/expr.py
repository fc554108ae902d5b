"""Expressions of the `w_i value` entries of a dflo .prm file.

dflo hands these strings to deal.II's FunctionParser (muparser) with the variables "x,y,t" for boundary
values (src/parameters.cc:441-477) and "x,y" for the initial condition (src/parameters.cc:481-493).  This is
a small recursive-descent parser of the muparser subset the shipped examples use (and a little more): numbers,
x y z t, pi e, + - * / ^, comparisons (value 1/0), && ||, unary minus, parentheses, if(c,a,b) and the usual
one- and two-argument functions.  An expression compiles to a closure evaluated with numpy on arrays.
"""
import math
import re

import numpy as np

_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*)|(<=|>=|==|!=|&&|\|\||[-+*/^()<>,?:]))")

_FUNC1 = {
    "sin": np.sin, "cos": np.cos, "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan,
    "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh, "exp": np.exp, "log": np.log, "ln": np.log,
    "log10": np.log10, "log2": np.log2, "sqrt": np.sqrt, "abs": np.abs, "sign": np.sign, "rint": np.rint,
    "floor": np.floor, "ceil": np.ceil, "int": np.trunc, "sec": lambda a: 1.0 / np.cos(a),
    "erfc": np.vectorize(math.erfc, otypes=[float]), "erf": np.vectorize(math.erf, otypes=[float]),
}
_FUNC2 = {"min": np.minimum, "max": np.maximum, "pow": np.power, "atan2": np.arctan2, "fmod": np.fmod}
_CONST = {"pi": math.pi, "Pi": math.pi, "PI": math.pi, "_pi": math.pi, "e": math.e, "_e": math.e}


class ExpressionError(ValueError):
    pass


def _tokenize(text):
    pos, out = 0, []
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ExpressionError("cannot parse %r at %r" % (text, text[pos:]))
        num, name, op = m.groups()
        out.append(("num", float(num)) if num is not None else (("name", name) if name is not None else ("op", op)))
        pos = m.end()
    out.append(("end", None))
    return out


class _Parser:
    """-> AST of nested tuples: ("num", v) ("var", name) ("neg", a) ("bin", op, a, b) ("call", name, [args]) ("sel", c, a, b)"""
    # precedence (low to high): ?:  ||  &&  == !=  < <= > >=  + -  * /  unary -  ^
    def __init__(self, text, variables):
        self.toks = _tokenize(text)
        self.i = 0
        self.variables = variables
        self.text = text

    def peek(self):
        return self.toks[self.i]

    def take(self, kind=None, val=None):
        t = self.toks[self.i]
        if (kind and t[0] != kind) or (val is not None and t[1] != val):
            raise ExpressionError("unexpected %r in %r" % (t[1], self.text))
        self.i += 1
        return t

    def is_op(self, *ops):
        t = self.peek()
        return t[0] == "op" and t[1] in ops

    def parse(self):
        f = self.ternary()
        self.take("end")
        return f

    def ternary(self):
        c = self.logic_or()
        if self.is_op("?"):
            self.take()
            a = self.ternary()
            self.take("op", ":")
            b = self.ternary()
            return ("sel", c, a, b)
        return c

    def _binary(self, sub, ops):
        f = sub()
        while self.is_op(*ops):
            op = self.take()[1]
            f = ("bin", op, f, sub())
        return f

    def logic_or(self):
        return self._binary(self.logic_and, ("||",))

    def logic_and(self):
        return self._binary(self.equality, ("&&",))

    def equality(self):
        return self._binary(self.relational, ("==", "!="))

    def relational(self):
        return self._binary(self.additive, ("<", "<=", ">", ">="))

    def additive(self):
        return self._binary(self.multiplicative, ("+", "-"))

    def multiplicative(self):
        return self._binary(self.unary, ("*", "/"))

    def unary(self):
        if self.is_op("-"):
            self.take()
            return ("neg", self.unary())
        if self.is_op("+"):
            self.take()
            return self.unary()
        return self.power()

    def power(self):
        base = self.atom()
        if self.is_op("^"):
            self.take()
            return ("bin", "^", base, self.unary())  # right associative, binds tighter than unary minus on its left
        return base

    def atom(self):
        kind, val = self.peek()
        if kind == "num":
            self.take()
            return ("num", val)
        if kind == "op" and val == "(":
            self.take()
            f = self.ternary()
            self.take("op", ")")
            return f
        if kind == "name":
            self.take()
            if self.is_op("("):
                self.take()
                args = [self.ternary()]
                while self.is_op(","):
                    self.take()
                    args.append(self.ternary())
                self.take("op", ")")
                if val == "if" and len(args) == 3:
                    return ("sel", args[0], args[1], args[2])
                if (val in _FUNC1 and len(args) == 1) or (val in _FUNC2 and len(args) == 2):
                    return ("call", val, args)
                raise ExpressionError("unknown function %s/%d in %r" % (val, len(args), self.text))
            if val in self.variables:
                return ("var", val)
            if val in _CONST:
                return ("num", _CONST[val])
            raise ExpressionError("unknown symbol %r in %r" % (val, self.text))
        raise ExpressionError("unexpected %r in %r" % (val, self.text))


_BIN = {
    "+": lambda a, b: a + b, "-": lambda a, b: a - b, "*": lambda a, b: a * b, "/": lambda a, b: a / b, "^": np.power,
    "<": lambda a, b: (a < b) * 1.0, "<=": lambda a, b: (a <= b) * 1.0, ">": lambda a, b: (a > b) * 1.0,
    ">=": lambda a, b: (a >= b) * 1.0, "==": lambda a, b: (a == b) * 1.0, "!=": lambda a, b: (a != b) * 1.0,
    "&&": lambda a, b: ((a != 0) & (b != 0)) * 1.0, "||": lambda a, b: ((a != 0) | (b != 0)) * 1.0,
}


def _evaluate(node, env):
    k = node[0]
    if k == "num":
        return node[1]
    if k == "var":
        return env[node[1]]
    if k == "neg":
        return -_evaluate(node[1], env)
    if k == "bin":
        return _BIN[node[1]](_evaluate(node[2], env), _evaluate(node[3], env))
    if k == "sel":
        return np.where(_evaluate(node[1], env) != 0, _evaluate(node[2], env), _evaluate(node[3], env))
    args = [_evaluate(a, env) for a in node[2]]
    return (_FUNC1[node[1]] if len(args) == 1 else _FUNC2[node[1]])(*args)


# ---- postfix programs for the device-side evaluation of boundary functions (dflo_hip_set_boundary_program);
#      the opcode numbers are part of the C ABI (include/dflo_hip.h, dflo_expr_op)
OP = {"const": 0, "x": 1, "y": 2, "t": 3, "neg": 4, "+": 5, "-": 6, "*": 7, "/": 8, "^": 9, "<": 10, "<=": 11, ">": 12,
      ">=": 13, "==": 14, "!=": 15, "&&": 16, "||": 17, "sel": 18, "sin": 19, "cos": 20, "tan": 21, "exp": 22, "log": 23,
      "sqrt": 24, "abs": 25, "min": 26, "max": 27, "atan2": 28, "tanh": 29, "sinh": 30, "cosh": 31, "asin": 32, "acos": 33,
      "atan": 34, "floor": 35, "ceil": 36, "sign": 37, "log10": 38, "pow": 9, "ln": 23, "erf": 39, "erfc": 40}
MAX_STACK = 16


def _emit(node, ops, consts):
    """Append the postfix form of `node` (opcode, operand index) and return the stack depth it needs."""
    k = node[0]
    if k == "num":
        ops.append((OP["const"], len(consts)))
        consts.append(float(node[1]))
        return 1
    if k == "var":
        if node[1] not in ("x", "y", "t"):
            raise ExpressionError("variable %r cannot be evaluated on the device" % node[1])
        ops.append((OP[node[1]], 0))
        return 1
    if k == "neg":
        d = _emit(node[1], ops, consts)
        ops.append((OP["neg"], 0))
        return d
    if k == "bin":
        da = _emit(node[2], ops, consts)
        db = _emit(node[3], ops, consts)
        ops.append((OP[node[1]], 0))
        return max(da, 1 + db)
    if k == "sel":
        dc = _emit(node[1], ops, consts)
        da = _emit(node[2], ops, consts)
        db = _emit(node[3], ops, consts)
        ops.append((OP["sel"], 0))
        return max(dc, 1 + da, 2 + db)
    if node[1] not in OP:
        raise ExpressionError("function %r cannot be evaluated on the device" % node[1])
    depth = 0
    for n, a in enumerate(node[2]):
        depth = max(depth, n + _emit(a, ops, consts))
    ops.append((OP[node[1]], 0))
    return depth


def _fold(node):
    """Evaluate the variable-free subtrees once, here (1.0/6.0, sqrt(3), ...): the device interprets what is left."""
    k = node[0]
    if k in ("num", "var"):
        return node
    if k == "neg":
        kids = [_fold(node[1])]
        out = ("neg", kids[0])
    elif k == "bin":
        kids = [_fold(node[2]), _fold(node[3])]
        out = ("bin", node[1], kids[0], kids[1])
    elif k == "sel":
        kids = [_fold(node[1]), _fold(node[2]), _fold(node[3])]
        out = ("sel",) + tuple(kids)
    else:
        kids = [_fold(a) for a in node[2]]
        out = ("call", node[1], kids)
    if all(c[0] == "num" for c in kids):
        with np.errstate(all="ignore"):
            return ("num", float(_evaluate(out, {})))
    return out


def compile_program(text, variables=("x", "y", "t")):
    """-> (ops int32 [n][2], consts float64 [m]) postfix program of `text` for the device evaluator."""
    ops, consts = [], []
    depth = _emit(_fold(_Parser(text, tuple(variables)).parse()), ops, consts)
    if depth > MAX_STACK:
        raise ExpressionError("expression %r needs an evaluation stack deeper than %d" % (text, MAX_STACK))
    return np.asarray(ops, dtype=np.int32).reshape(-1, 2), np.asarray(consts, dtype=np.float64)


def run_program(ops, consts, x, y, t):
    """Host interpreter of a postfix program (mirror of the device kernel; used to test the compiler)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    names = {v: k for k, v in OP.items() if k not in ("pow", "ln")}
    st = []
    for op, arg in np.asarray(ops).reshape(-1, 2):
        n = names[int(op)]
        if n == "const":
            st.append(np.full(x.shape, consts[arg]))
        elif n in ("x", "y", "t"):
            st.append({"x": x, "y": y, "t": np.full(x.shape, float(t))}[n])
        elif n == "neg":
            st.append(-st.pop())
        elif n in _BIN:
            b, a = st.pop(), st.pop()
            st.append(_BIN[n](a, b))
        elif n == "sel":
            b, a, c = st.pop(), st.pop(), st.pop()
            st.append(np.where(c != 0, a, b))
        elif n in _FUNC2:
            b, a = st.pop(), st.pop()
            st.append(_FUNC2[n](a, b))
        else:
            st.append(_FUNC1[n](st.pop()))
    assert len(st) == 1
    return st[0]


def compile_expression(text, variables=("x", "y", "t")):
    """-> f(**values) evaluating `text`; values are numpy arrays or scalars, the result broadcasts over them."""
    ast = _Parser(text, tuple(variables)).parse()

    def evaluate(**values):
        shape = np.broadcast(*[np.asarray(values[v]) for v in variables if v in values]).shape if values else ()
        r = _evaluate(ast, {v: np.asarray(values.get(v, 0.0), dtype=np.float64) for v in variables})
        return np.broadcast_to(np.asarray(r, dtype=np.float64), shape).copy()

    return evaluate


class VectorFunction:
    """The 4-component FunctionParser of a boundary / initial-condition subsection (components w_0..w_3 =
    x-momentum, y-momentum, density, energy)."""

    def __init__(self, expressions, variables=("x", "y", "t")):
        self.expressions = list(expressions)
        self.variables = tuple(variables)
        self._fn = [compile_expression(e, variables) for e in expressions]
        self.time_dependent = "t" in variables and any(re.search(r"(?<![A-Za-z_0-9])t(?![A-Za-z_0-9(])", e) for e in expressions)

    def __call__(self, x, y, t=0.0):
        kw = {"x": x, "y": y}
        if "t" in self.variables:
            kw["t"] = t
        return [f(**kw) for f in self._fn]
