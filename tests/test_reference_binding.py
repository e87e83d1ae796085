"""The reference-side binding of the drop-in (pathtracer_amd/csrc/Makefile target `pathtracer`):
the reference's UNCHANGED src/c/main.c + src/c/lua_wrapper.c must compile against include/renderer.h once that
header stands in the place of src/c/renderer.h.  Container-only (needs /root/reference; skipped on the GPU box).

Nothing of the reference is copied: the staging directory holds symlinks, made in tmp_path at test time.  The image
has no Lua, so the three Lua headers are minimal test-only stand-ins declaring what lua_wrapper.c uses — enough for
-fsyntax-only, which is all that is claimed here (types, prototypes, struct members of renderer.h as the binding
uses them).  `gcc -H` lists every header opened: ours must be there, the reference's renderer.h must not."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

LUA_H = r'''
#ifndef lua_h
#define lua_h
#include <stddef.h>
typedef struct lua_State lua_State;
typedef double lua_Number;
typedef long long lua_Integer;
typedef int (*lua_CFunction)(lua_State *L);
#define LUA_OK 0
#define LUA_MULTRET (-1)
void *lua_touserdata(lua_State *L, int idx);
void lua_setfield(lua_State *L, int idx, const char *k);
int lua_getfield(lua_State *L, int idx, const char *k);
void lua_settop(lua_State *L, int idx);
#define lua_pop(L, n) lua_settop(L, -(n) - 1)
void lua_pushnumber(lua_State *L, lua_Number n);
void lua_pushinteger(lua_State *L, lua_Integer n);
void lua_pushlightuserdata(lua_State *L, void *p);
const char *lua_tolstring(lua_State *L, int idx, size_t *len);
#define lua_tostring(L, i) lua_tolstring(L, (i), NULL)
void lua_createtable(lua_State *L, int narr, int nrec);
#define lua_newtable(L) lua_createtable(L, 0, 0)
int lua_pcallk(lua_State *L, int nargs, int nresults, int errfunc, long ctx, void *k);
#define lua_pcall(L, n, r, f) lua_pcallk(L, (n), (r), (f), 0, NULL)
void lua_setglobal(lua_State *L, const char *name);
int lua_getglobal(lua_State *L, const char *name);
void lua_close(lua_State *L);
#endif
'''
LAUXLIB_H = r'''
#ifndef lauxlib_h
#define lauxlib_h
#include "lua.h"
typedef struct luaL_Reg { const char *name; lua_CFunction func; } luaL_Reg;
lua_Number luaL_checknumber(lua_State *L, int arg);
lua_Integer luaL_checkinteger(lua_State *L, int arg);
const char *luaL_checklstring(lua_State *L, int arg, size_t *l);
#define luaL_checkstring(L, n) luaL_checklstring(L, (n), NULL)
void luaL_setfuncs(lua_State *L, const luaL_Reg *l, int nup);
lua_State *luaL_newstate(void);
int luaL_loadfilex(lua_State *L, const char *filename, const char *mode);
#define luaL_loadfile(L, f) luaL_loadfilex(L, f, NULL)
#endif
'''
LUALIB_H = r'''
#ifndef lualib_h
#define lualib_h
#include "lua.h"
void luaL_openlibs(lua_State *L);
#endif
'''


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src/c/lua_wrapper.c")), reason="needs the reference checkout")
def test_unchanged_reference_front_end_compiles_against_our_renderer_h(tmp_path):
    stage = tmp_path / "stage"
    (stage / "src" / "c").mkdir(parents=True)
    for f in ("main.c", "lua_wrapper.c", "lua_wrapper.h"):
        os.symlink(os.path.join(REF, "src/c", f), stage / "src" / "c" / f)
    os.symlink(os.path.join(ROOT, "include", "renderer.h"), stage / "src" / "c" / "renderer.h")
    os.symlink(os.path.join(REF, "include"), stage / "include")
    lua = tmp_path / "lua_stub"
    lua.mkdir()
    for name, text in (("lua.h", LUA_H), ("lauxlib.h", LAUXLIB_H), ("lualib.h", LUALIB_H)):
        (lua / name).write_text(text)
    opened = []
    for src in ("main.c", "lua_wrapper.c"):
        r = subprocess.run(["gcc", "-fsyntax-only", "-H", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", str(lua),
                            str(stage / "src" / "c" / src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        opened += [ln.lstrip(". ").strip() for ln in r.stderr.splitlines() if ln.startswith(".")]
    real = {os.path.realpath(p) for p in opened}
    assert os.path.realpath(os.path.join(ROOT, "include", "renderer.h")) in real, "our renderer.h was not the one opened"
    assert os.path.realpath(os.path.join(ROOT, "include", "CL", "cl.h")) in real      # main.c:7's unused include
    ref_hdrs = {p for p in real if p.startswith(REF + "/src/")}
    assert ref_hdrs == {os.path.join(REF, "src/c/lua_wrapper.h")}, ref_hdrs           # and never the reference's renderer.h
    assert not any(p.endswith("/CL/cl.h") and not p.startswith(ROOT) for p in real), "a system OpenCL header was pulled in"


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src/c/lua_wrapper.c")), reason="needs the reference checkout")
def test_compiled_where_it_lies_the_reference_header_wins(tmp_path):
    """The negative: from src/c/ itself, -I/-idirafter cannot beat the quote-include (what round 1's recipe got wrong)."""
    lua = tmp_path / "lua_stub"
    lua.mkdir()
    for name, text in (("lua.h", LUA_H), ("lauxlib.h", LAUXLIB_H), ("lualib.h", LUALIB_H)):
        (lua / name).write_text(text)
    r = subprocess.run(["gcc", "-fsyntax-only", "-H", "-I", os.path.join(ROOT, "include"), "-I", str(lua), "-idirafter",
                        os.path.join(REF, "src/c"), os.path.join(REF, "src/c/lua_wrapper.c")], capture_output=True, text=True)
    opened = {os.path.realpath(ln.lstrip(". ").strip()) for ln in r.stderr.splitlines() if ln.startswith(".")}
    assert os.path.join(REF, "src/c/renderer.h") in opened
