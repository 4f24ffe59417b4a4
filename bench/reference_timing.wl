(* reference_timing.wl -- times the TRUE reference (ssmit1986/BayesianInference, Wolfram Language)
   on cfg 1 of BASELINE.json (N = 512, d = 1, squared-exponential kernel, fp64) so that a user who
   owns a Wolfram kernel can put a measured number next to bench.py's "cpu_baseline" line.

   NOT run in the build containers or on the GPU box (no Wolfram engine there; SURVEY.md §8c/§8d);
   shipped unrun.  Usage:

       wolframscript -file bench/reference_timing.wl /path/to/BayesianInference [N]

   The synthetic data are the same counter-based splitmix64 stream as
   bayesianinference_amd/synthetic.py (seed 20250905), restated here with exact integer arithmetic,
   so "same inputs" holds bit for bit; the kernel text is the SE form of SURVEY.md §8d.  The script
   prints the log marginal likelihood at the timing hyper-parameters (compare with
   tests/golden/f1_*.npz / `python bench.py`), the definition time (symbolic covariance + Compile,
   BayesianGaussianProcess.wl:45-61) and the per-evaluation time of the closure the sampler calls
   (BayesianGaussianProcess.wl:297-305). *)

args = Rest @ $ScriptCommandLine;
refDir = If[Length[args] >= 1, args[[1]], "."];
n = If[Length[args] >= 2, ToExpression @ args[[2]], 512];

PacletDirectoryLoad[refDir];
Needs["BayesianInference`"];

(* ---- synthetic.py restated: mix64(seed + stream*STREAM + (idx+1)*GOLDEN), u = (bits >> 11) 2^-53 ---- *)
mask = 2^64 - 1;
golden = 16^^9E3779B97F4A7C15; streamMul = 16^^D1B54A32D192ED03;
m1 = 16^^BF58476D1CE4E5B9; m2 = 16^^94D049BB133111EB;
mix64[z0_] := Module[{z = z0},
	z = BitAnd[BitXor[z, BitShiftRight[z, 30]] m1, mask];
	z = BitAnd[BitXor[z, BitShiftRight[z, 27]] m2, mask];
	BitXor[z, BitShiftRight[z, 31]]
];
uniform[stream_, start_, count_] := Table[
	N[BitShiftRight[mix64[BitAnd[20250905 + stream streamMul + (i + 1) golden, mask]], 11]/2^53],
	{i, start, start + count - 1}
];
normal[stream_, start_, count_] := With[{u = Partition[uniform[stream, 2 start, 2 count], 2]},
	Sqrt[-2. Log[1. - u[[All, 1]]]] Cos[2. Pi u[[All, 2]]]
];

d = 1;
xs = Partition[2. uniform[0, 0, n d] - 1., d];
ys = Sin[2. (xs . (1./Range[d]))] + 0.1 normal[1, 0, n];

(* ---- the reference's own public entry point (BayesianGaussianProcess.wl:228-234) ---- *)
tDefine = First @ AbsoluteTiming[
	obj = defineGaussianProcess[
		xs -> Transpose[{ys}],
		Function[{p, q}, sf^2 Exp[-Total[(p - q)^2]/(2 ell^2)]],
		Function[sn^2],
		Function[0],
		{{ell, 0.05, 5.}, {sf, 0.05, 5.}, {sn, 0.01, 1.}},
		ProductDistribution[UniformDistribution[{0.05, 5.}], UniformDistribution[{0.05, 5.}], UniformDistribution[{0.01, 1.}]]
	];
];
loglik = obj["LogLikelihoodFunction"];
theta = {0.3, 1.0, 0.1};

value = loglik[theta];
reps = 10;
tEval = First @ AbsoluteTiming[Do[loglik[theta], {reps}]]/reps;

Print["N = ", n, "  d = ", d, "  kernel = SE  theta = ", theta];
Print["log marginal likelihood = ", NumberForm[value, 17]];
Print["defineGaussianProcess (symbolic K + Compile): ", tDefine, " s"];
Print["per evaluation: ", tEval, " s  => ", 1/tEval, " evals/s (reference, ", $ProcessorCount, " cores)"];
