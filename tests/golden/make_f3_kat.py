"""Known-answer values for PosOrnKeypointDistFunct::diff (reference src/system/PosOrnKeypointDistFunct.cpp:13-35), derived BY HAND from the
reference's text -- not computed by the oracle, the device code or any restatement of them.  Writes tests/golden/f3_deadzone_kat.json.

Construction.  PosOrnKeypoint::diff (PosOrnKeypoint.cpp:24-45) gives the plain residual
    r = [ p* - p ;  -2 H(q*) log_{q*}(q) ].
For unit quaternions with q = delta (x) q*, delta = (cos(theta/2), sin(theta/2) n), |n| = 1, 0 <= theta < pi (sd.h:48-82: the chord
y - (b.y) b has length sin(theta/2), the geodesic distance is acos(cos(theta/2)) = theta/2, no antipodal fold because b.y > 0):
    log_{q*}(q) = (theta/2) (0, n) (x) q*,      H(q*) w = vec(w (x) conj(q*))  (sd.h:23-27, row by row)
    =>  -2 H(q*) log_{q*}(q) = -theta n.
So a case is fixed by the position residual r_p = p* - p, the rotation vector theta n, the radius and the three thresholds, and the plain
residual is r = (r_p, -theta n) in closed form.  The dead zone (:16-32) then is
    position:     |r_p| <= radius -> 0,  else r_p / |r_p| * (|r_p| - radius)
    orientation:  |r_i| <= thresh_i -> 0,  else r_i - sign(r_i) thresh_i          (equality falls INSIDE the zone in both tests)
The expected vectors below are these expressions evaluated by hand on numbers chosen to make that possible (3-4-5 triangles).

`abs` at :26 is unqualified.  Which overload it names depends on the headers in scope: with only <cmath>/<cstdlib> g++ picks int abs(int)
and TRUNCATES (abs(0.7) == 0); the reference includes <eigen3/Eigen/Dense>, whose Core header pulls in <emmintrin.h> -> <mm_malloc.h> ->
<stdlib.h> (libstdc++'s C++ wrapper, `using std::abs`) on every x86-64 build, and then abs(double) is the floating-point overload.
Both facts were checked with g++ 11 in this container (two-line programs; see DESIGN.md "Oracle").  Case "abs_overload" is the one
whose answer differs between the two readings; the floating-point one is the reference's.
"""
import json
import os

CASES = [
    # r_p = (0.3, 0.4, 0): |r_p| = 0.5.  theta n = 0.5 * (0.6, -0.8, 0) = (0.3, -0.4, 0)  ->  r_orn = (-0.3, 0.4, 0)
    dict(name="outside_ball_and_thresholds", r_pos=[0.3, 0.4, 0.0], theta=0.5, axis=[0.6, -0.8, 0.0], pos_radius=0.1, orn_thresh=[0.1, 0.1, 0.1],
         plain=[0.3, 0.4, 0.0, -0.3, 0.4, 0.0],
         # position: (0.3, 0.4, 0) / 0.5 * (0.5 - 0.1) = (0.24, 0.32, 0);  orientation: -0.3 + 0.1, 0.4 - 0.1, |0| <= 0.1 -> 0
         expect=[0.24, 0.32, 0.0, -0.2, 0.3, 0.0],
         # e' diag(1,1,1,.1,.1,.1) e = 0.0576 + 0.1024 + 0.1 (0.04 + 0.09) = 0.173
         cost_Q=[1, 1, 1, 0.1, 0.1, 0.1], cost=0.173),
    dict(name="inside_ball", r_pos=[0.3, 0.4, 0.0], theta=0.5, axis=[0.6, -0.8, 0.0], pos_radius=0.6, orn_thresh=[0.0, 0.0, 0.0],
         plain=[0.3, 0.4, 0.0, -0.3, 0.4, 0.0],
         # |r_p| = 0.5 <= 0.6 -> 0;  thresholds 0: |-0.3| > 0 -> -0.3 + 0 ; |0.4| > 0 -> 0.4 ; |0| <= 0 -> 0
         expect=[0.0, 0.0, 0.0, -0.3, 0.4, 0.0], cost_Q=[1, 1, 1, 1, 1, 1], cost=0.25),
    dict(name="radius_zero", r_pos=[0.0, -0.6, 0.8], theta=0.25, axis=[0.0, 0.0, 1.0], pos_radius=0.0, orn_thresh=[0.5, 0.5, 0.1],
         plain=[0.0, -0.6, 0.8, 0.0, 0.0, -0.25],
         # |r_p| = 1 > 0 -> r_p * (1 - 0) = r_p;  orientation: 0, 0 inside; -0.25 + 0.1 = -0.15
         expect=[0.0, -0.6, 0.8, 0.0, 0.0, -0.15], cost_Q=[1, 1, 1, 1, 1, 1], cost=1.0225),
    dict(name="on_the_thresholds", r_pos=[0.0, 0.0, 0.25], theta=0.5, axis=[0.6, -0.8, 0.0], pos_radius=0.25, orn_thresh=[0.5, 0.25, 0.0],
         plain=[0.0, 0.0, 0.25, -0.3, 0.4, 0.0],
         # |r_p| = 0.25 <= 0.25 -> 0 (equality is inside);  |-0.3| <= 0.5 -> 0;  0.4 - 0.25 = 0.15;  0
         # (0.5 * 0.6 and 0.5 * -0.8 are not exactly 0.3 / -0.4 in binary: these two thresholds are kept away from equality on purpose)
         expect=[0.0, 0.0, 0.0, 0.0, 0.15, 0.0], cost_Q=[1, 1, 1, 1, 1, 1], cost=0.0225),
    dict(name="negative_components", r_pos=[-0.8, 0.0, -0.6], theta=1.0, axis=[0.0, 0.6, 0.8], pos_radius=0.5, orn_thresh=[0.1, 0.1, 0.3],
         plain=[-0.8, 0.0, -0.6, 0.0, -0.6, -0.8],
         # |r_p| = 1 -> r_p * 0.5 = (-0.4, 0, -0.3);  0 -> 0;  -0.6 + 0.1 = -0.5;  -0.8 + 0.3 = -0.5
         expect=[-0.4, 0.0, -0.3, 0.0, -0.5, -0.5], cost_Q=[1, 1, 1, 1, 1, 1], cost=0.75),
    dict(name="abs_overload", r_pos=[0.0, 0.0, 0.0], theta=0.75, axis=[-1.0, 0.0, 0.0], pos_radius=0.0, orn_thresh=[0.5, 0.5, 0.5],
         plain=[0.0, 0.0, 0.0, 0.75, 0.0, 0.0],
         # floating-point abs: 0.75 > 0.5 -> 0.75 - 0.5 = 0.25   (int abs would give abs(0.75) == 0 <= 0.5 -> 0)
         # position: |r_p| = 0 <= 0 -> 0
         expect=[0.0, 0.0, 0.0, 0.25, 0.0, 0.0], cost_Q=[1, 1, 1, 1, 1, 1], cost=0.0625),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f3_deadzone_kat.json")
    json.dump(dict(source="hand-derived from PosOrnKeypointDistFunct.cpp:13-35 and PosOrnKeypoint.cpp:24-45; see make_f3_kat.py", cases=CASES), open(out, "w"), indent=1)
    print(out)
