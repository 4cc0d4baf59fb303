"""Example models shared by the GPU tests and by the build step that compiles their device code ahead
of time (so that no GPU test waits for the compiler)."""
import numpy as np

from ilqr_amd import systems


class DraggedCar(systems.Car):
    """``class MyCar(Car)`` with its own dynamics, written in the reference's idiom (systems.py:33-57):
    the solver must call THESE methods (ilqr.py:62, :87-88), i.e. record and compile them."""
    drag = 0.05
    DRAGS = (0.05, 0.2)   # the values the tests solve with

    def model_f(self, x, u):
        xn = np.array(systems.Car.model_f(self, x, u), dtype=object if x.dtype == object else float)
        xn[2] = xn[2] - self.drag * x[2] * self.dt
        return xn

    def compute_df_dx(self, x, u):
        J = np.array(systems.Car.compute_df_dx(self, x, u), dtype=object if x.dtype == object else float)
        J[2, 2] = J[2, 2] - self.drag * self.dt
        return J
