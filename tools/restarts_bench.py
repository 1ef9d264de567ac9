# scratch: wall clock of the drivers' n_init loop on the C2 / C3 stand-ins, data resident vs
# uploaded per fit; prints one line per configuration
import sys, os, time, warnings
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import convex_dim_red as cdr
from convex_dim_red import _backend
from oracle import aa_oracle as orc
warnings.simplefilter("ignore")
n_init = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def c2():
    n, p, k = 1610, 25000, 5
    rng = np.random.RandomState(0); B = rng.standard_normal((k, p)); Zt = orc.right_stochastic_matrix((n, k), rng) ** 4; Zt /= Zt.sum(axis=1, keepdims=True)
    return Zt.dot(B) + 0.05 * rng.standard_normal((n, p)), k
def c3():
    n, p, k = 22280, 167, 10
    rng = np.random.RandomState(0); W0 = rng.standard_normal((p, k)); Zt = orc.right_stochastic_matrix((n, k), rng)
    return Zt.dot(W0.T) + 0.1 * rng.standard_normal((n, p)), k

def loop(make, X):
    shared = np.random.RandomState(0); costs = []; iters = 0
    t0 = time.perf_counter()
    for _ in range(n_init):
        m = make(shared); m.fit_transform(X); costs.append(m.cost); iters += m.n_iter + 1
    return time.perf_counter() - t0, costs, iters

for name, (X, k), make in (
        ("C2 AA k=5 1610x25000 tol 1e-4", c2(), lambda rs: cdr.ArchetypalAnalysis(5, init="random", tolerance=1e-4, max_iterations=10000, random_state=rs, dictionary_solver_kwargs=dict(max_iterations=1))),
        ("C3 GPNH k=10 22280x167 lambda=0 tol 1e-6", c3(), lambda rs: cdr.GPNHConvexCoding(10, lambda_W=0, init="random", tolerance=1e-6, max_iterations=10000, random_state=rs, stopping_criterion="rel_delta_f", weights_solver_kwargs=dict(max_iterations=1)))):
    _backend.release_device_cache()
    loop(make, X[:64])                                   # warm the library
    _backend.release_device_cache()
    t_res, c_res, it = loop(make, X)
    os.environ["CONVEX_DIM_RED_CACHE"] = "0"; _backend.release_device_cache()
    t_up, c_up, _ = loop(make, X)
    os.environ.pop("CONVEX_DIM_RED_CACHE")
    print("%s: n_init=%d, %d outer iterations: %.2f s resident (%.0f it/s), %.2f s per-fit upload; identical costs: %s"
          % (name, n_init, it, t_res, it / t_res, t_up, c_res == c_up), flush=True)
    for jobs in (1, 2, 4):                               # the same restarts, several fits at a time (threads)
        shared = np.random.RandomState(0)
        t0 = time.perf_counter()
        models, best = cdr.fit_restarts(lambda: make(shared), X, n_init, n_jobs=jobs, side_by_side=False)
        t = time.perf_counter() - t0
        print("   fit_restarts n_jobs=%d: %.2f s (%.0f it/s), identical costs: %s, best restart %d"
              % (jobs, t, it / t, [m.cost for m in models] == c_res, best), flush=True)
    if name.startswith("C2"):                            # AA: restarts side by side in one set of arrays
        shared = np.random.RandomState(0)
        t0 = time.perf_counter()
        models, best = cdr.fit_restarts(lambda: make(shared), X, n_init)
        t = time.perf_counter() - t0
        print("   fit_restarts side by side, %d at a time: %.3f s (%.0f it/s) = %.2fx the sequential loop, identical costs: %s, best restart %d"
              % (32 // k, t, it / t, t_res / t, [m.cost for m in models] == c_res, best), flush=True)
    if name.startswith("C3"):                            # GPNH: restarts side by side in one set of arrays
        for slots in (3, 6):
            shared = np.random.RandomState(0)
            t0 = time.perf_counter()
            models, best = cdr.fit_restarts(lambda: make(shared), X, n_init, n_slots=slots)
            t = time.perf_counter() - t0
            print("   fit_restarts side by side, %d slots: %.3f s (%.0f it/s) = %.2fx the sequential loop, identical costs: %s, best restart %d"
                  % (slots, t, it / t, t_res / t, [m.cost for m in models] == c_res, best), flush=True)
