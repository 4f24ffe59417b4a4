(* make_reference_golden.wl -- ONE COMMAND that pins parity against the REAL reference.

   Runs the reference's own code -- defineGaussianProcess (BayesianGaussianProcess.wl:228-330), the closure it
   installs (:297-305), its "InverseCovarianceFunction" (:308, :130-141) and predictFromGaussianProcess in its
   direct form (:378-394 -> predictFromGaussianProcessInternal :396-422) -- on the inputs of the committed golden
   fixtures (tests/golden/reference_inputs.json, written by oracle/export_reference_inputs.py) and writes
   tests/golden/reference_<case>.json.  tests/test_oracle.py::test_oracle_matches_reference_fixture and
   tests/test_gpu_reference.py consume those files when present (1e-10 / 1e-8 relative) and skip with the reason
   "parity unpinned" when absent.

   It CANNOT run in the build containers or on the GPU box (no Wolfram kernel; SURVEY.md §8c): it is shipped
   unrun, and until someone with Mathematica 14+ runs it parity stays "unpinned" (DESIGN.md §6).

       wolframscript -file oracle/make_reference_golden.wl /path/to/BayesianInference [maxN]

   maxN (default 600) skips cases with more training points: the reference builds a symbolic N x N covariance
   and Compiles it (BayesianGaussianProcess.wl:45-61), minutes at N ~ 500. *)

args = Rest @ $ScriptCommandLine;
refDir = If[Length[args] >= 1, args[[1]], "."];
maxN = If[Length[args] >= 2, ToExpression @ args[[2]], 600];
here = DirectoryName[$InputFileName];
goldenDir = FileNameJoin[{ParentDirectory[here], "tests", "golden"}];

PacletDirectoryLoad[refDir];
Needs["BayesianInference`"];

inputs = Import[FileNameJoin[{goldenDir, "reference_inputs.json"}], "RawJSON"];

(* the named kernels in the WL form SURVEY.md §8d fixes (the same text GPHIP`hipKernelFunction uses), with the
   hyper-parameters as SYMBOLS so that the reference's varsToParamVector substitution (:50-51) is exercised *)
lsyms[d_] := Table[Symbol["ell" <> ToString[j]], {j, d}];
kernelExpr["se", d_] := Function[{p, q}, sf^2 Exp[-Total[(p - q)^2]/(2 ell1^2)]];
kernelExpr["se_ard", d_] := With[{ls = lsyms[d]}, Function[{p, q}, sf^2 Exp[-Total[((p - q)/ls)^2]/2]]];
kernelExpr["matern52", d_] := Function[{p, q},
	With[{s = Sqrt[Total[(p - q)^2]]/ell1}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]];
kernelExpr["matern52_ard", d_] := With[{ls = lsyms[d]}, Function[{p, q},
	With[{s = Sqrt[Total[((p - q)/ls)^2]]}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]];
nl["se" | "matern52", d_] := 1;
nl["se_ard" | "matern52_ard", d_] := d;

runCase[case_] := Module[{
	name = case["name"], kernel = case["kernel"], constMean = case["mean"] === "const",
	x = N @ case["X"], y = N @ case["y"], xs = N @ case["Xs"], thetas = N @ case["thetas"],
	d, vars, specs, obj, loglik, invCov, ll, logdet, quad, preds, numericRules, kf, nf, mf, t0
},
	d = Dimensions[x][[2]];
	vars = Join[Take[lsyms[d], nl[kernel, d]], {sf, sn}, If[constMean, {mu}, {}]];
	specs = {#, -1000., 1000.}& /@ vars;        (* a wide box: the definition-time smoke test only needs finite values *)
	t0 = AbsoluteTime[];
	obj = defineGaussianProcess[
		x -> Transpose[{y}],
		kernelExpr[kernel, d],
		Function[sn^2],
		If[constMean, Function[mu], Function[0]],
		specs,
		ProductDistribution @@ (UniformDistribution[{-1000., 1000.}]& /@ vars)
	];
	If[ !MatchQ[obj, _inferenceObject] || FailureQ[obj[[1]]],
		Print[name, ": defineGaussianProcess failed"]; Return[$Failed]
	];
	loglik = obj["LogLikelihoodFunction"];
	invCov = obj["GaussianProcessData", "ModelFunctions", "InverseCovarianceFunction"];
	ll = loglik /@ thetas;
	(* LogDet and the quadratic form separately, through the reference's own matrixInverseAndDet; a singular K
	   Throws the sentinel with tag "MatInv" (:133) *)
	{logdet, quad} = Transpose @ Map[
		Function[th,
			Catch[
				With[{ic = invCov[th], r = y - If[constMean, Last[th], 0.]},
					{ic["LogDet"], r . ic["Inverse"][r]}
				],
				"MatInv", {Missing["MatInv"], Missing["MatInv"]}&
			]
		],
		thetas
	];
	(* direct prediction form (:378-394) with the hyper-parameters substituted numerically *)
	preds = Table[
		numericRules = Thread[vars -> thetas[[i]]];
		kf = kernelExpr[kernel, d] /. numericRules;
		nf = Function[sn^2] /. numericRules;
		mf = If[constMean, Function[mu] /. numericRules, Function[0]];
		With[{res = predictFromGaussianProcess[x -> Transpose[{y}], xs, kf, nf, mf]},
			{Mean /@ Values[res], StandardDeviation /@ Values[res], Keys[res]}
		],
		{i, Min[case["npred"], Length[thetas]]}
	];
	Export[
		FileNameJoin[{goldenDir, "reference_" <> name <> ".json"}],
		<|
			"name" -> name, "kernel" -> kernel, "mean" -> case["mean"],
			"generator" -> "oracle/make_reference_golden.wl",
			"wolfram_version" -> $Version, "machine_log_zero" -> $MachineLogZero,
			"seconds" -> AbsoluteTime[] - t0,
			"thetas" -> thetas,
			"loglik" -> ll,
			"loglik_is_sentinel" -> (# === $MachineLogZero & /@ ll),
			"logdet" -> (logdet /. _Missing -> Null), "quad" -> (quad /. _Missing -> Null),
			"pred_points" -> If[preds === {}, {}, preds[[1, 3]]],
			"pred_mu" -> preds[[All, 1]], "pred_sd" -> preds[[All, 2]]
		|>,
		"RawJSON"
	];
	Print[name, ": ", Length[thetas], " thetas, ", Length[preds], " predictions, ", Round[AbsoluteTime[] - t0, 0.1], " s"];
];

Scan[
	If[ Length[#["X"]] <= maxN, runCase[#], Print[#["name"], ": skipped (N > ", maxN, ")"]]&,
	inputs["cases"]
];
